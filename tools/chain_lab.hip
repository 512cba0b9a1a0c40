// Lab: what does the 16-pivot chain of the diagonal-block factor cost, and which part of it?  One wavefront runs
// chain16 (gpr_amd/csrc/chol.hip) REPS times on a 16 x 16 block held in registers; s_memtime around the loop.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/chain_lab.hip gpr_amd/_build/common.cpp.o gpr_amd/_build/mfma_gemm.hip.o -o build/chain_lab
#include <cstdio>
#include <vector>
#include "../gpr_amd/csrc/chol.hip"
using namespace gprhip;

template <int VAR>
__global__ void lab_kernel(const double* blk, double* out, unsigned long long* ticks, int reps) {
  const int cc = threadIdx.x & 15;
  double a0[16];
  for (int r = 0; r < 16; ++r) a0[r] = blk[r * 16 + cc];
  double acc = 0.0;
  const unsigned long long t0 = clock64();
  for (int it = 0; it < reps; ++it) {
    double a[16], y[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      a[r] = a0[r] + acc * 1e-300;  // (loop-carried, so that the repetitions do not overlap)
      y[r] = (r == cc) ? 1.0 : 0.0;
    }
    int badq = 0;
    double myrp = 0.0;
    chain16(a, y, cc, 0, badq, myrp);
    acc += a[15] + y[15] + myrp + badq;
  }
  const unsigned long long t1 = clock64();
  if (threadIdx.x == 0) ticks[0] = t1 - t0;
  out[threadIdx.x] = acc;
}

int main() {
  std::vector<double> G(256), A(256);
  for (auto& v : G) v = (double)rand() / RAND_MAX - 0.5;
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      double s = 0;
      for (int k = 0; k < 16; ++k) s += G[i * 16 + k] * G[j * 16 + k];
      A[i * 16 + j] = s + (i == j ? 1.0 : 0.0);
    }
  double *dA, *dO;
  unsigned long long* dT;
  hipMalloc(&dA, 256 * 8); hipMalloc(&dO, 64 * 8); hipMalloc(&dT, 8);
  hipMemcpy(dA, A.data(), 256 * 8, hipMemcpyHostToDevice);
  const int reps = 2000;
  for (int rep = 0; rep < 3; ++rep) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(lab_kernel<0>, dim3(1), dim3(64), 0, 0, dA, dO, dT, reps);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long t; hipMemcpy(&t, dT, 8, hipMemcpyDeviceToHost);
    printf("chain16: %.1f ticks of s_memtime per chain, %.3f us per chain (%d repetitions, %.3f ms)\n", (double)t / reps, ms * 1e3 / reps, reps, ms);
  }
  return 0;
}
