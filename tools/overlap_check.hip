// Feasibility lab for launching column groups of the V = K U^-1 product while the factorisation of K_m still runs
// (DESIGN.md section 13, item 5): how long does the blocked factorisation + carried inverse (m = 2048) take
//   (a) alone,
//   (b) while a long engine launch (a 131072 x 2048 x 2048 triangular product) runs on a second stream,
//   (c) the same with the second stream confined by a CU mask (hipExtStreamCreateWithCUMask) that leaves CUs free,
// and how much the engine launch slows down under the mask.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../gpr_amd/csrc/kernels.h"
#include "../gpr_amd/csrc/mfma_gemm.h"
using namespace gprhip;

int main() {
  gemm_init();
  const int m = 2048, rows = 131072;
  std::vector<double> B((size_t)m * 64), S((size_t)m * m, 0.0);
  for (auto& v : B) v = (double)rand() / RAND_MAX - 0.5;
  for (int i = 0; i < m; ++i)
    for (int j = i; j < m; ++j) {
      double s = 0;
      for (int k = 0; k < 64; ++k) s += B[(size_t)i * 64 + k] * B[(size_t)j * 64 + k];
      S[(size_t)i * m + j] = s + (i == j ? 1e-2 * (1 + i % 7) : 0.0);
    }
  double *dS, *dV, *dY, *dX, *dA, *dC, *dU;
  int* dJ;
  hipMalloc(&dS, (size_t)m * m * 8); hipMalloc(&dV, (size_t)(m / 128) * 128 * 128 * 8); hipMalloc(&dJ, 8);
  hipMalloc(&dY, (size_t)m * m * 8); hipMalloc(&dX, (size_t)m * m * 8); hipMalloc(&dU, (size_t)m * m * 8);
  hipMalloc(&dA, (size_t)rows * m * 8); hipMalloc(&dC, (size_t)rows * m * 8);
  hipMemset(dJ, 0, 8);
  hipMemset(dA, 0, (size_t)rows * m * 8);
  hipMemset(dU, 0, (size_t)m * m * 8);
  hipStream_t s1, s2, s3;
  hipStreamCreate(&s1);
  hipStreamCreate(&s2);
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  hipEvent_t e0, e1, g0, g1;
  hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&g0); hipEventCreate(&g1);
  auto gemm = [&](hipStream_t st) {
    GemmArgs g;
    g.A = dA; g.lda = m; g.B = dU; g.ldb = m; g.C = dC; g.ldc = m;
    g.M = rows; g.N = m; g.K = m; g.tri = TRI_KHI_BN; g.order = 3;
    launch_gemm(OP_NN, g, st);
  };
  auto potrf = [&]() {
    hipMemcpyAsync(dS, S.data(), (size_t)m * m * 8, hipMemcpyHostToDevice, s1);
    hipStreamSynchronize(s1);
  };
  auto run = [&](const char* name, hipStream_t gs) {
    float best_p = 1e9, best_g = 0;
    for (int rep = 0; rep < 4; ++rep) {
      potrf();
      if (gs) {
        hipEventRecord(g0, gs);
        gemm(gs);
        hipEventRecord(g1, gs);
        // let the engine launch fill the chip before the factorisation starts
        hipStreamSynchronize(s1);
        for (volatile int spin = 0; spin < 2000000; ++spin) {}
      }
      hipEventRecord(e0, s1);
      potrf_upper_blocked(s1, dS, m, dV, dJ, dY, dX);
      hipEventRecord(e1, s1);
      hipEventSynchronize(e1);
      if (gs) hipEventSynchronize(g1);
      float ms, gms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      if (gs) hipEventElapsedTime(&gms, g0, g1);
      if (ms < best_p) { best_p = ms; best_g = gms; }
    }
    printf("%-64s factorisation+inverse %.3f ms   engine launch %.3f ms\n", name, best_p, best_g);
  };
  {  // the engine launch alone, unmasked
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(g0, s2); gemm(s2); hipEventRecord(g1, s2); hipEventSynchronize(g1);
      float ms; hipEventElapsedTime(&ms, g0, g1); best = std::min(best, ms);
    }
    printf("engine launch alone (all %d CUs): %.3f ms\n", cus, best);
  }
  run("(a) factorisation alone", nullptr);
  run("(b) beside an engine launch on an unmasked stream", s2);
  for (int freecu : {16, 32, 64}) {
    // enable all CUs but `freecu` of them; the driver spreads mask bits over the XCDs
    std::vector<uint32_t> mask((cus + 31) / 32, 0);
    for (int i = 0; i < cus - freecu; ++i) mask[i / 32] |= 1u << (i % 32);
    hipError_t e = hipExtStreamCreateWithCUMask(&s3, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) {
      printf("hipExtStreamCreateWithCUMask failed: %s\n", hipGetErrorString(e));
      break;
    }
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(g0, s3); gemm(s3); hipEventRecord(g1, s3); hipEventSynchronize(g1);
      float ms; hipEventElapsedTime(&ms, g0, g1); best = std::min(best, ms);
    }
    char name[128];
    snprintf(name, sizeof name, "(c) beside an engine launch masked to %d of %d CUs", cus - freecu, cus);
    printf("engine launch alone on the masked stream (%d CUs): %.3f ms\n", cus - freecu, best);
    run(name, s3);
    hipStreamDestroy(s3);
  }
  return 0;
}
