// Standalone correctness + timing harness for the fp64 MFMA engine (runs on the GPU box).
#include <vector>
#include <cmath>
#include <cstdlib>
#include <algorithm>
#include "../gpr_amd/csrc/mfma_gemm.h"
using namespace gprhip;

__global__ void naive(int op, const double* A, int64_t lda, const double* B, int64_t ldb, double* C,
                      int64_t ldc, int M, int N, int K, const double* sk) {
  int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (j >= N) return;
  double s = 0;
  for (int k = 0; k < K; ++k) {
    double a = op == OP_TN ? A[(int64_t)k * lda + i] * (sk ? sk[k] : 1.0) : A[(int64_t)i * lda + k];
    double b = op == OP_NT ? B[(int64_t)j * ldb + k] : B[(int64_t)k * ldb + j];
    s += a * b;
  }
  C[(int64_t)i * ldc + j] = s;
}

static double frand() { return (double)rand() / RAND_MAX * 2 - 1; }

int check(GemmOp op, int M, int N, int K, int tri, bool scale) {
  int64_t ar = op == OP_TN ? K : M, ac = op == OP_TN ? M : K;
  int64_t br = op == OP_NT ? N : K, bc = op == OP_NT ? K : N;
  std::vector<double> hA(ar * ac), hB(br * bc), hs(K);
  for (auto& v : hA) v = frand();
  for (auto& v : hB) v = frand();
  for (auto& v : hs) v = frand();
  // zero the parts the triangular policies assume zero
  if (tri == TRI_KHI_BN) for (int k = 0; k < K; ++k) for (int j = 0; j < N; ++j) if (k > j) hB[(int64_t)k * bc + j] = 0;
  if (tri == TRI_KLO_BN) for (int j = 0; j < N; ++j) for (int k = 0; k < K; ++k) if (k < j) hB[(int64_t)j * bc + k] = 0;
  if (tri == TRI_KLO_BM) for (int i = 0; i < M; ++i) for (int k = 0; k < K; ++k) if (k < i) hA[(int64_t)i * ac + k] = 0;
  if (tri == TRI_KLO_MAX) {
    for (int i = 0; i < M; ++i) for (int k = 0; k < K; ++k) if (k < i) hA[(int64_t)i * ac + k] = 0;
    for (int j = 0; j < N; ++j) for (int k = 0; k < K; ++k) if (k < j) hB[(int64_t)j * bc + k] = 0;
  }
  double *dA, *dB, *dC, *dR, *ds;
  hipMalloc(&dA, hA.size() * 8); hipMalloc(&dB, hB.size() * 8); hipMalloc(&ds, K * 8);
  hipMalloc(&dC, (int64_t)M * N * 8); hipMalloc(&dR, (int64_t)M * N * 8);
  hipMemcpy(dA, hA.data(), hA.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dB, hB.data(), hB.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(ds, hs.data(), K * 8, hipMemcpyHostToDevice);
  hipMemset(dC, 0, (int64_t)M * N * 8);
  GemmArgs g; g.A = dA; g.lda = ac; g.B = dB; g.ldb = bc; g.C = dC; g.ldc = N; g.M = M; g.N = N; g.K = K;
  g.order = getenv("ORD") ? atoi(getenv("ORD")) : 0;
  g.tri = tri; g.scale_k = scale ? ds : nullptr;
  double *drp = nullptr, *drd = nullptr, *dvec = nullptr;
  const int nparts = 2 * (N / 128);
  if (getenv("RP")) {  // row reductions from the epilogue (sum of squares and dot with a vector), as the V / Q' products use them
    hipMalloc(&drp, (int64_t)nparts * M * 8); hipMalloc(&drd, (int64_t)nparts * M * 8); hipMalloc(&dvec, N * 8);
    hipMemset(drp, 0xff, (int64_t)nparts * M * 8); hipMemset(drd, 0xff, (int64_t)nparts * M * 8);
    hipMemcpy(dvec, hs.data(), std::min(N, K) * 8, hipMemcpyHostToDevice);
    g.rp_sumsq = drp; g.rp_dot = drd; g.rp_vec = dvec;
  }
  launch_gemm(op, g, 0);
  naive<<<dim3((N + 255) / 256, M), 256>>>(op, dA, ac, dB, bc, dR, N, M, N, K, scale ? ds : nullptr);
  std::vector<double> hC((int64_t)M * N), hR((int64_t)M * N);
  hipMemcpy(hC.data(), dC, hC.size() * 8, hipMemcpyDeviceToHost);
  hipMemcpy(hR.data(), dR, hR.size() * 8, hipMemcpyDeviceToHost);
  double maxerr = 0;
  for (size_t i = 0; i < hC.size(); ++i) maxerr = std::max(maxerr, fabs(hC[i] - hR[i]));
  if (getenv("WHERE") && maxerr > 1e-10 * K) {  // which tiles are wrong (count of bad elements per column tile; first bad row panels)
    const int nbn = N / 128, nbm = M / 128;
    std::vector<int> cnt(nbn, 0), firstbm(nbn, -1), tiles(nbn, 0);
    for (int bm = 0; bm < nbm; ++bm)
      for (int bn = 0; bn < nbn; ++bn) {
        int bad = 0;
        for (int r = 0; r < 128; ++r)
          for (int q = 0; q < 128; ++q) {
            const int64_t ix = (int64_t)(bm * 128 + r) * N + bn * 128 + q;
            if (fabs(hC[ix] - hR[ix]) > 1e-10 * K) ++bad;
          }
        if (bad && firstbm[bn] < 0 && bn < 2) {
          printf("    tile (%d,%d) bad rows:", bm, bn);
          for (int r = 0; r < 128; ++r) { int c = 0; for (int q = 0; q < 128; ++q) { const int64_t ix = (int64_t)(bm * 128 + r) * N + bn * 128 + q; if (fabs(hC[ix] - hR[ix]) > 1e-10 * K) ++c; } if (c) printf(" %d(%d)", r, c); }
          printf("\n    bad cols:");
          for (int q = 0; q < 128; ++q) { int c = 0; for (int r = 0; r < 128; ++r) { const int64_t ix = (int64_t)(bm * 128 + r) * N + bn * 128 + q; if (fabs(hC[ix] - hR[ix]) > 1e-10 * K) ++c; } if (c) printf(" %d(%d)", q, c); }
          printf("\n");
        }
        if (bad) { cnt[bn] += bad; ++tiles[bn]; if (firstbm[bn] < 0) firstbm[bn] = bm; }
      }
    for (int bn = 0; bn < nbn; ++bn) printf("    col tile %d: %d bad tiles, %d bad elements, first bad row panel %d\n", bn, tiles[bn], cnt[bn], firstbm[bn]);
  }
  if (drp && N <= K) {
    std::vector<double> hp((int64_t)nparts * M), hd((int64_t)nparts * M);
    hipMemcpy(hp.data(), drp, hp.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(hd.data(), drd, hd.size() * 8, hipMemcpyDeviceToHost);
    double e2 = 0, ed = 0;
    for (int i = 0; i < M; ++i) {
      double s2 = 0, sd = 0, r2 = 0, rd = 0;
      for (int q = 0; q < nparts; ++q) { s2 += hp[(int64_t)q * M + i]; sd += hd[(int64_t)q * M + i]; }
      for (int j = 0; j < N; ++j) { const double v = hR[(int64_t)i * N + j]; r2 += v * v; rd += v * hs[j]; }
      e2 = std::max(e2, fabs(s2 - r2) / (1.0 + r2)); ed = std::max(ed, fabs(sd - rd) / (1.0 + fabs(rd)));
    }
    printf("  row partials: sumsq err %.3e dot err %.3e %s\n", e2, ed, (e2 < 1e-10 && ed < 1e-10) ? "OK" : "FAIL");
    printf("  C maxerr %.3e\n", maxerr);
    if (!(e2 < 1e-10 && ed < 1e-10)) maxerr = 1e300;
    hipFree(drp); hipFree(drd); hipFree(dvec);
  }
  printf("check op=%d M=%d N=%d K=%d tri=%d scale=%d maxerr=%.3e %s\n", op, M, N, K, tri, (int)scale, maxerr,
         maxerr < 1e-10 * K ? "OK" : "FAIL");
  hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dR); hipFree(ds);
  return maxerr < 1e-10 * K ? 0 : 1;
}

void timeit(GemmOp op, int M, int N, int K, int tri, int upper, int kslices, const char* name, bool scale = false) {
  int64_t ar = op == OP_TN ? K : M, ac = op == OP_TN ? M : K;
  int64_t br = op == OP_NT ? N : K, bc = op == OP_NT ? K : N;
  double *dA, *dB, *dC;
  hipMalloc(&dA, ar * ac * 8); hipMalloc(&dB, br * bc * 8);
  hipMalloc(&dC, (int64_t)M * N * 8 * std::max(1, kslices));
  std::vector<double> h(std::max(ar * ac, br * bc));
  for (size_t i = 0; i < h.size(); ++i) h[i] = (double)((i * 2654435761u) & 0xffff) / 65536.0 - 0.5;
  hipMemcpy(dA, h.data(), ar * ac * 8, hipMemcpyHostToDevice);
  hipMemcpy(dB, h.data(), br * bc * 8, hipMemcpyHostToDevice);
  hipMemset(dC, 0, (int64_t)M * N * 8 * std::max(1, kslices));
  double* dB0 = dB;
  if (op == OP_TN && upper && M == N) dB = dA;  // SYRK-shaped: one operand (the engine's diagonal-tile variant applies)
  double* dS = nullptr; if (scale) { hipMalloc(&dS, (int64_t)K * 8); hipMemcpy(dS, h.data(), (int64_t)K * 8, hipMemcpyHostToDevice); }
  GemmArgs g; g.scale_k = dS; g.A = dA; g.lda = ac; g.B = dB; g.ldb = bc; g.C = dC; g.ldc = N; g.M = M; g.N = N; g.K = K;
  g.tri = tri; g.upper_only = upper; g.kslices = kslices; g.order = getenv("ORD") ? atoi(getenv("ORD")) : 0; g.slice_stride = (int64_t)M * N; g.lab_skip = getenv("SKIP") ? atoi(getenv("SKIP")) : 0;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) launch_gemm(op, g, 0);
  hipEventRecord(e0, 0);
  int reps = 5;
  for (int i = 0; i < reps; ++i) launch_gemm(op, g, 0);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  double flops = 2.0 * M * N * (double)K;
  if (tri != TRI_NONE) flops *= 0.5;
  if (upper) flops *= 0.5;
  printf("time %-28s M=%d N=%d K=%d tri=%d upper=%d ks=%d : %.3f ms  %.1f TFLOP/s (useful)\n", name, M, N, K, tri,
         upper, kslices, ms, flops / ms * 1e-9);
  hipFree(dA); hipFree(dB0); hipFree(dC);
}

template <typename T>
__global__ void naive_t(int op, const T* A, int64_t lda, const T* B, int64_t ldb, double* C, int64_t ldc, int M,
                        int N, int K, const double* sk) {
  int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (j >= N) return;
  double s = 0;
  for (int k = 0; k < K; ++k) {
    double a = op == OP_TN ? (double)A[(int64_t)k * lda + i] * (sk ? sk[k] : 1.0) : (double)A[(int64_t)i * lda + k];
    double b = op == OP_NT ? (double)B[(int64_t)j * ldb + k] : (double)B[(int64_t)k * ldb + j];
    s += a * b;
  }
  C[(int64_t)i * ldc + j] = s;
}

int check_f32(GemmOp op, int M, int N, int K, int tri, bool scale, bool epi) {
  int64_t ar = op == OP_TN ? K : M, ac = op == OP_TN ? M : K;
  int64_t br = op == OP_NT ? N : K, bc = op == OP_NT ? K : N;
  std::vector<float> hA(ar * ac), hB(br * bc), hM((int64_t)M * N);
  std::vector<double> hs(std::max(K, M)), hcol(N);
  std::vector<float> hsf(std::max(K, M));
  for (auto& v : hA) v = (float)frand();
  for (auto& v : hB) v = (float)frand();
  for (auto& v : hM) v = (float)frand();
  for (auto& v : hs) v = frand();
  for (size_t i = 0; i < hs.size(); ++i) { hsf[i] = (float)hs[i]; hs[i] = hsf[i]; }
  for (auto& v : hcol) v = frand();
  if (tri == TRI_KHI_BN) for (int k = 0; k < K; ++k) for (int j = 0; j < N; ++j) if (k > j) hB[(int64_t)k * bc + j] = 0;
  if (tri == TRI_KLO_BN) for (int j = 0; j < N; ++j) for (int k = 0; k < K; ++k) if (k < j) hB[(int64_t)j * bc + k] = 0;
  float *dA, *dB, *dC, *dM, *dsf; double *dR, *ds, *dcol;
  hipMalloc(&dA, hA.size() * 4); hipMalloc(&dB, hB.size() * 4); hipMalloc(&dC, (int64_t)M * N * 4); hipMalloc(&dM, (int64_t)M * N * 4);
  hipMalloc(&dR, (int64_t)M * N * 8); hipMalloc(&ds, hs.size() * 8); hipMalloc(&dcol, N * 8); hipMalloc(&dsf, hs.size() * 4);
  hipMemcpy(dsf, hsf.data(), hsf.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dM, hM.data(), hM.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(ds, hs.data(), hs.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dcol, hcol.data(), N * 8, hipMemcpyHostToDevice);
  hipMemset(dC, 0, (int64_t)M * N * 4);
  GemmArgsF g; g.A = dA; g.lda = ac; g.B = dB; g.ldb = bc; g.C = dC; g.ldc = N; g.M = M; g.N = N; g.K = K; g.tri = tri;
  g.order = getenv("ORD") ? atoi(getenv("ORD")) : 0;
  g.scale_k = scale ? dsf : nullptr;
  if (epi) { g.epi_rows_a = ds; g.epi_rows_b = ds; g.epi_rows_c = ds; g.epi_col = dcol; g.epi_mat = dM; g.epi_ldm = N; }
  launch_gemm(op, g, 0);
  naive_t<float><<<dim3((N + 255) / 256, M), 256>>>(op, dA, ac, dB, bc, dR, N, M, N, K, scale ? ds : nullptr);
  std::vector<float> hC((int64_t)M * N); std::vector<double> hR((int64_t)M * N);
  hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(hR.data(), dR, hR.size() * 8, hipMemcpyDeviceToHost);
  double maxerr = 0;
  for (int i = 0; i < M; ++i) for (int j = 0; j < N; ++j) {
    double ref = hR[(int64_t)i * N + j];
    if (epi) ref = hs[i] * ref - hs[i] * hM[(int64_t)i * N + j] - hs[i] * hcol[j];
    maxerr = std::max(maxerr, fabs(hC[(int64_t)i * N + j] - ref));
  }
  bool ok = maxerr < 2e-6 * K;
  printf("check f32 op=%d M=%d N=%d K=%d tri=%d scale=%d epi=%d maxerr=%.3e %s\n", op, M, N, K, tri, (int)scale, (int)epi, maxerr, ok ? "OK" : "FAIL");
  hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dM); hipFree(dR); hipFree(ds); hipFree(dcol);
  return ok ? 0 : 1;
}

// Two-phase NT item with the M-less fused epilogue (the X product of the gradient pass):
//   C[i][j] = ra[i] * (A B^T - (num[i] / den[i]) A2 B2^T)[i][j] - rc[i] * cv[j],   rows with den == 0 take no A2 term
template <typename T>
int check_two_phase(int M, int N, int K) {
  std::vector<T> hA((int64_t)M * K), hA2((int64_t)M * K), hB((int64_t)N * K), hB2((int64_t)N * K);
  std::vector<double> ra(M), rc(M), num(M), den(M), cv(N);
  for (auto& v : hA) v = (T)frand();
  for (auto& v : hA2) v = (T)frand();
  for (auto& v : hB) v = (T)frand();
  for (auto& v : hB2) v = (T)frand();
  for (int j = 0; j < N; ++j) for (int k = 0; k < K; ++k) if (k < j) hB[(int64_t)j * K + k] = hB2[(int64_t)j * K + k] = 0;
  for (int i = 0; i < M; ++i) { ra[i] = frand(); rc[i] = frand(); num[i] = frand(); den[i] = (i % 37 == 5) ? 0.0 : 0.5 + fabs(frand()); }
  for (auto& v : cv) v = frand();
  T *dA, *dA2, *dB, *dB2, *dC; double *dra, *drc, *dnum, *dden, *dcv, *dR1, *dR2;
  hipMalloc(&dA, hA.size() * sizeof(T)); hipMalloc(&dA2, hA.size() * sizeof(T)); hipMalloc(&dB, hB.size() * sizeof(T));
  hipMalloc(&dB2, hB.size() * sizeof(T)); hipMalloc(&dC, (int64_t)M * N * sizeof(T));
  hipMalloc(&dra, M * 8); hipMalloc(&drc, M * 8); hipMalloc(&dnum, M * 8); hipMalloc(&dden, M * 8); hipMalloc(&dcv, N * 8);
  hipMalloc(&dR1, (int64_t)M * N * 8); hipMalloc(&dR2, (int64_t)M * N * 8);
  hipMemcpy(dA, hA.data(), hA.size() * sizeof(T), hipMemcpyHostToDevice); hipMemcpy(dA2, hA2.data(), hA.size() * sizeof(T), hipMemcpyHostToDevice);
  hipMemcpy(dB, hB.data(), hB.size() * sizeof(T), hipMemcpyHostToDevice); hipMemcpy(dB2, hB2.data(), hB.size() * sizeof(T), hipMemcpyHostToDevice);
  hipMemcpy(dra, ra.data(), M * 8, hipMemcpyHostToDevice); hipMemcpy(drc, rc.data(), M * 8, hipMemcpyHostToDevice);
  hipMemcpy(dnum, num.data(), M * 8, hipMemcpyHostToDevice); hipMemcpy(dden, den.data(), M * 8, hipMemcpyHostToDevice);
  hipMemcpy(dcv, cv.data(), N * 8, hipMemcpyHostToDevice);
  hipMemset(dC, 0xff, (int64_t)M * N * sizeof(T));
  GemmArgsT<T> g; g.A = dA; g.lda = K; g.B = dB; g.ldb = K; g.C = dC; g.ldc = N; g.M = M; g.N = N; g.K = K; g.tri = TRI_KLO_BN;
  g.order = getenv("ORD") ? atoi(getenv("ORD")) : 0;
  g.A2 = dA2; g.B2 = dB2; g.mid_num = dnum; g.mid_den = dden;
  g.epi_rows_a = dra; g.epi_rows_c = drc; g.epi_col = dcv;
  launch_gemm(OP_NT, g, 0);
  naive_t<T><<<dim3((N + 255) / 256, M), 256>>>(OP_NT, dA, K, dB, K, dR1, N, M, N, K, nullptr);
  naive_t<T><<<dim3((N + 255) / 256, M), 256>>>(OP_NT, dA2, K, dB2, K, dR2, N, M, N, K, nullptr);
  std::vector<T> hC((int64_t)M * N); std::vector<double> h1((int64_t)M * N), h2((int64_t)M * N);
  hipMemcpy(hC.data(), dC, hC.size() * sizeof(T), hipMemcpyDeviceToHost);
  hipMemcpy(h1.data(), dR1, h1.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(h2.data(), dR2, h2.size() * 8, hipMemcpyDeviceToHost);
  double maxerr = 0;
  for (int i = 0; i < M; ++i) for (int j = 0; j < N; ++j) {
    const double f = den[i] != 0.0 ? num[i] / den[i] : 0.0;
    const double ref = ra[i] * (h1[(int64_t)i * N + j] - f * h2[(int64_t)i * N + j]) - rc[i] * cv[j];
    maxerr = std::max(maxerr, fabs((double)hC[(int64_t)i * N + j] - ref));
  }
  const bool ok = maxerr < (sizeof(T) == 8 ? 1e-10 : 4e-6) * K;
  printf("check two-phase %s M=%d N=%d K=%d maxerr=%.3e %s\n", sizeof(T) == 8 ? "f64" : "f32", M, N, K, maxerr, ok ? "OK" : "FAIL");
  hipFree(dA); hipFree(dA2); hipFree(dB); hipFree(dB2); hipFree(dC); hipFree(dra); hipFree(drc); hipFree(dnum); hipFree(dden);
  hipFree(dcv); hipFree(dR1); hipFree(dR2);
  return ok ? 0 : 1;
}

void timeit_f32(GemmOp op, int M, int N, int K, int tri, int upper, int kslices, const char* name) {
  int64_t ar = op == OP_TN ? K : M, ac = op == OP_TN ? M : K;
  int64_t br = op == OP_NT ? N : K, bc = op == OP_NT ? K : N;
  float *dA, *dB, *dC;
  hipMalloc(&dA, ar * ac * 4); hipMalloc(&dB, br * bc * 4); hipMalloc(&dC, (int64_t)M * N * 4 * std::max(1, kslices));
  std::vector<float> h(std::max(ar * ac, br * bc));
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) & 0xffff) / 65536.0f - 0.5f;
  hipMemcpy(dA, h.data(), ar * ac * 4, hipMemcpyHostToDevice); hipMemcpy(dB, h.data(), br * bc * 4, hipMemcpyHostToDevice);
  GemmArgsF g; g.A = dA; g.lda = ac; g.B = dB; g.ldb = bc; g.C = dC; g.ldc = N; g.M = M; g.N = N; g.K = K;
  g.tri = tri; g.upper_only = upper; g.kslices = kslices; g.slice_stride = (int64_t)M * N;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) launch_gemm(op, g, 0);
  hipEventRecord(e0, 0);
  int reps = 5;
  for (int i = 0; i < reps; ++i) launch_gemm(op, g, 0);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  double flops = 2.0 * M * N * (double)K; if (tri != TRI_NONE) flops *= 0.5; if (upper) flops *= 0.5;
  printf("time f32 %-24s M=%d N=%d K=%d tri=%d upper=%d ks=%d : %.3f ms  %.1f TFLOP/s (useful)\n", name, M, N, K, tri, upper, kslices, ms, flops / ms * 1e-9);
  hipFree(dA); hipFree(dB); hipFree(dC);
}


// weighted SYRK-shaped split-K launch with column sums on the diagonal tiles (the pass-1 accumulation)
template <typename T>
int check_syrk_cs(int N, int K, int ks) {
  std::vector<T> hA((int64_t)K * N), hw(K), hw2(K);
  for (auto& v : hA) v = (T)frand();
  for (auto& v : hw) v = (T)(0.5 + 0.5 * frand());
  for (auto& v : hw2) v = (T)frand();
  T *dA, *dw, *dw2, *dC; double* dcs;
  hipMalloc(&dA, hA.size() * sizeof(T)); hipMalloc(&dw, K * sizeof(T)); hipMalloc(&dw2, K * sizeof(T));
  hipMalloc(&dC, (int64_t)ks * N * N * sizeof(T)); hipMalloc(&dcs, (int64_t)ks * N * 8);
  hipMemcpy(dA, hA.data(), hA.size() * sizeof(T), hipMemcpyHostToDevice);
  hipMemcpy(dw, hw.data(), K * sizeof(T), hipMemcpyHostToDevice);
  hipMemcpy(dw2, hw2.data(), K * sizeof(T), hipMemcpyHostToDevice);
  hipMemset(dC, 0, (int64_t)ks * N * N * sizeof(T)); hipMemset(dcs, 0, (int64_t)ks * N * 8);
  GemmArgsT<T> g; g.A = dA; g.lda = N; g.B = dA; g.ldb = N; g.C = dC; g.ldc = N; g.M = N; g.N = N; g.K = K;
  g.scale_k = dw; g.upper_only = 1; g.kslices = ks; g.slice_stride = (int64_t)N * N; g.cs_w = dw2; g.cs_out = dcs;
  launch_gemm(OP_TN, g, 0);
  std::vector<T> hC((int64_t)ks * N * N); std::vector<double> hcs((int64_t)ks * N);
  hipMemcpy(hC.data(), dC, hC.size() * sizeof(T), hipMemcpyDeviceToHost);
  hipMemcpy(hcs.data(), dcs, hcs.size() * 8, hipMemcpyDeviceToHost);
  double maxerr = 0, maxcs = 0;
  for (int i = 0; i < N; ++i)
    for (int j = i / 128 * 128; j < N; ++j) {   // upper tiles
      double ref = 0, got = 0;
      for (int z = 0; z < ks; ++z) got += hC[(int64_t)z * N * N + (int64_t)i * N + j];
      if (j < i / 16 * 16) {  // strictly-lower 16 x 16 sub-tile of a diagonal tile: not computed, stays zero
        maxerr = std::max(maxerr, fabs(got));
        continue;
      }
      for (int k = 0; k < K; ++k) ref += (double)hA[(int64_t)k * N + i] * hw[k] * hA[(int64_t)k * N + j];
      maxerr = std::max(maxerr, fabs(ref - got));
    }
  for (int c = 0; c < N; ++c) {
    double ref = 0, got = 0;
    for (int k = 0; k < K; ++k) ref += (double)hA[(int64_t)k * N + c] * hw2[k];
    for (int z = 0; z < ks; ++z) got += hcs[(int64_t)z * N + c];
    maxcs = std::max(maxcs, fabs(ref - got));
  }
  const double tol = (sizeof(T) == 8 ? 1e-10 : 2e-5) * K;
  bool ok = maxerr < tol && maxcs < tol;
  printf("check syrk+colsums %s N=%d K=%d ks=%d maxerr=%.3e colsum err=%.3e %s\n", sizeof(T) == 8 ? "f64" : "f32", N, K, ks, maxerr, maxcs, ok ? "OK" : "FAIL");
  hipFree(dA); hipFree(dw); hipFree(dw2); hipFree(dC); hipFree(dcs);
  return ok ? 0 : 1;
}

// Register-only MFMA stream: the matrix-pipe ceiling at the clock the chip sustains under this load
// (no LDS, no global traffic).  16 independent accumulators per wavefront, 4 wavefronts per block.
typedef double d4_t __attribute__((ext_vector_type(4)));
template <int WPS>
__global__ __launch_bounds__(256, WPS) void mfma_peak_kernel(double* out, int iters, double a0, double b0) {
  d4_t acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = (d4_t){0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i & 7], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678) out[0] = s;
}
template <int WPS>
void peak(int blocks_per_cu, const char* name) {
  double* d; hipMalloc(&d, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;  // 20000 x 16 MFMA x 64 cyc = 20.5 M cycles per wave alone
  const int grid = 256 * blocks_per_cu;
  hipLaunchKernelGGL(mfma_peak_kernel<WPS>, dim3(grid), dim3(256), 0, 0, d, 2000, 1.0, 1.0);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(mfma_peak_kernel<WPS>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0, 1.0);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)grid * 4 * iters * 16 * 2048.0;
  printf("peak %-20s grid=%d : %.3f ms  %.1f TFLOP/s\n", name, grid, ms, flops / ms * 1e-9);
  hipFree(d);
}

// TS=1: per-workgroup phase time stamps of one triangular chunk product (written to TS_OUT as raw uint64[wg][8])
void timestamps(GemmOp op, int M, int N, int K, int tri, const char* path) {
  int64_t ar = M, ac = K, br = op == OP_NT ? N : K, bc = op == OP_NT ? K : N;
  double *dA, *dB, *dC;
  hipMalloc(&dA, ar * ac * 8); hipMalloc(&dB, br * bc * 8); hipMalloc(&dC, (int64_t)M * N * 8);
  {  // realistic operand values: the chip's clock under fp64 MFMA load depends on the data
    std::vector<double> h(std::max(ar * ac, br * bc));
    for (size_t i = 0; i < h.size(); ++i) h[i] = (double)((i * 2654435761u) & 0xffff) / 65536.0 - 0.5;
    hipMemcpy(dB, h.data(), br * bc * 8, hipMemcpyHostToDevice);
    const int data = getenv("DATA") ? atoi(getenv("DATA")) : 0;
    if (data == 1)       // covariance-like A: exp(-1/2 chi^2_8)-distributed values in (0, 1]
      for (size_t i = 0; i < (size_t)(ar * ac); ++i) {
        double c2 = 0;
        for (int k = 0; k < 8; ++k) { const double g = (double)(((i * 8 + k) * 2246822519u) & 0xffff) / 65536.0 - 0.5; c2 += 12.0 * g * g; }
        h[i] = exp(-0.5 * c2);
      }
    else if (data == 2)  // full-mantissa random values of mixed magnitude
      for (size_t i = 0; i < (size_t)(ar * ac); ++i) h[i] = ((double)rand() / RAND_MAX - 0.5) * exp(10.0 * ((double)rand() / RAND_MAX - 0.5));
    hipMemcpy(dA, h.data(), ar * ac * 8, hipMemcpyHostToDevice);
  }
  const int nwg = (M / 128) * (N / 128);
  unsigned long long* dts; hipMalloc(&dts, (size_t)nwg * 128); hipMemset(dts, 0, (size_t)nwg * 128);
  GemmArgs g; g.A = dA; g.lda = ac; g.B = dB; g.ldb = bc; g.C = dC; g.ldc = N; g.M = M; g.N = N; g.K = K;
  g.tri = tri; g.order = getenv("ORD") ? atoi(getenv("ORD")) : 3;
  for (int i = 0; i < 2; ++i) launch_gemm(op, g, 0);
  g.lab_ts = dts;
  launch_gemm(op, g, 0);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h((size_t)nwg * 16);
  hipMemcpy(h.data(), dts, h.size() * 8, hipMemcpyDeviceToHost);
  FILE* f = fopen(path, "wb"); fwrite(h.data(), 8, h.size(), f); fclose(f);
  {
    unsigned long long lo = ~0ull, hi = 0;
    for (int w = 0; w < nwg; ++w) if (h[(size_t)w * 16 + 10]) { lo = std::min(lo, h[(size_t)w * 16 + 10]); for (int q = 0; q < 6; ++q) hi = std::max(hi, h[(size_t)w * 16 + q]); }
    printf("timestamps: %d workgroup records -> %s; first start to last end %.1f us\n", nwg, path, (hi - lo) / 100.0);
  }
  hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dts);
}

// TS_SYRK=1 [WS=1]: per-workgroup time stamps of the 1M-row weighted SYRK launch of the headline shape, through the plain
// weighted kernel (gemm_f64_tn_w) or, with WS=1, the column-sum kernel (gemm_f64_tn_ws): how far apart the workgroups of
// an XCD's residency rounds start, and how long their k-loops last (the pass-2 SYRK finding)
template <typename T>
__global__ void fill_kernel(T* x, size_t n, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    unsigned h = (unsigned)(i * 2654435761u) ^ seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    x[i] = (T)((double)(h & 0xffffff) / 16777216.0 - 0.5);
  }
}
template <typename T>
void timestamps_syrk(bool ws) {
  const bool f64 = sizeof(T) == 8;
  const int N = getenv("NCOLS") ? atoi(getenv("NCOLS")) : 2048, KS = getenv("KS") ? atoi(getenv("KS")) : 112;
  const int64_t K = 1000064;
  T *dV, *dW, *dC;
  double* dCS;
  hipMalloc(&dV, K * N * sizeof(T)); hipMalloc(&dW, K * sizeof(T)); hipMalloc(&dC, (int64_t)KS * N * N * sizeof(T));
  hipMalloc(&dCS, (int64_t)KS * N * 8);
  hipLaunchKernelGGL(fill_kernel<T>, dim3(4096), dim3(256), 0, 0, dV, (size_t)(K * N), 1u);
  hipLaunchKernelGGL(fill_kernel<T>, dim3(256), dim3(256), 0, 0, dW, (size_t)K, 7u);
  GemmArgsT<T> g;
  g.A = dV; g.lda = N; g.B = dV; g.ldb = N; g.C = dC; g.ldc = N; g.M = N; g.N = N; g.K = (int)K;
  g.scale_k = dW; g.upper_only = 1; g.kslices = KS; g.slice_stride = (int64_t)N * N;
  if (ws) { g.cs_w = dW; g.cs_out = dCS; }
  const int nbn = N / 128, tiles = nbn * (nbn + 1) / 2;
  const int dsl = gemm_syrk_diag_slices(KS, f64, ws);
  const int nwg = 8 * ((tiles - nbn) * (KS / 8) + nbn * ((dsl + 7) / 8));
  unsigned long long* dts; hipMalloc(&dts, (size_t)nwg * 128); hipMemset(dts, 0, (size_t)nwg * 128);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) launch_gemm(OP_TN, g, 0);
  hipEventRecord(e0, 0); launch_gemm(OP_TN, g, 0); hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  g.lab_ts = dts;
  launch_gemm(OP_TN, g, 0);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h((size_t)nwg * 16);
  hipMemcpy(h.data(), dts, h.size() * 8, hipMemcpyDeviceToHost);
  printf("%s %s N=%d: %d workgroups, %d + %d slices, launch %.3f ms (without stamps)\n", f64 ? "f64" : "f32", ws ? "tn_ws" : "tn_w", N,
         nwg, KS, dsl, ms);
  // per XCD, in dispatch order (blockIdx = 8 q + x): k-loop start of position q, in groups of 64 positions
  unsigned long long t0 = ~0ull;
  for (int w = 0; w < nwg; ++w) if (h[(size_t)w * 16 + 1]) t0 = std::min(t0, h[(size_t)w * 16 + 10]);
  const int nq = nwg / 8, nrounds = (nq + 63) / 64;
  for (int x = 0; x < 8; x += 7) {  // first and last XCD
    printf("XCD %d: round (64 positions): first loop start [us], spread of loop starts [us], mean / min / max loop length [us]\n", x);
    for (int r = 0; r < nrounds; r += std::max(1, nrounds / 24)) {
      double lo = 1e30, hi = 0, sum = 0, dmin = 1e30, dmax = 0; int cnt = 0;
      for (int q = r * 64; q < std::min(nq, r * 64 + 64); ++q) {
        const unsigned long long* rec = &h[(size_t)(8 * q + x) * 16];
        if (!rec[1]) continue;
        const double st = (rec[1] - t0) / 100.0, len = (rec[2] - rec[1]) / 100.0;
        lo = std::min(lo, st); hi = std::max(hi, st); sum += len; dmin = std::min(dmin, len); dmax = std::max(dmax, len); ++cnt;
      }
      if (cnt) {
        printf("  round %3d: start %9.1f  spread %7.1f   loop %7.1f / %7.1f / %7.1f  (%d items)", r, lo, hi - lo, sum / cnt, dmin, dmax, cnt);
        // sorted loop starts relative to the earliest, every eighth: two tight groups or a continuum?
        std::vector<double> st;
        for (int q = r * 64; q < std::min(nq, r * 64 + 64); ++q) {
          const unsigned long long* rec = &h[(size_t)(8 * q + x) * 16];
          if (rec[1]) st.push_back((rec[1] - t0) / 100.0 - lo);
        }
        std::sort(st.begin(), st.end());
        printf("   starts:");
        for (size_t i = 0; i < st.size(); i += 7) printf(" %.0f", st[i]);
        printf("\n");
      }
    }
  }
  hipFree(dV); hipFree(dW); hipFree(dC); hipFree(dCS); hipFree(dts);
}

int main() {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  gemm_init();
  if (getenv("TS_SYRK")) {
    if (getenv("F32")) timestamps_syrk<float>(getenv("WS") != nullptr);
    else timestamps_syrk<double>(getenv("WS") != nullptr);
    return 0;
  }
  if (getenv("TS")) {
    const int tm = getenv("TS_M") ? atoi(getenv("TS_M")) : 131072, tn = getenv("TS_N") ? atoi(getenv("TS_N")) : 2048;  // rows (a multiple of 128), m
    timestamps(OP_NN, tm, tn, tn, TRI_KHI_BN, getenv("TS_OUT") ? getenv("TS_OUT") : "ts_nn.bin");
    return 0;
  }
  if (getenv("PEAK")) {
    peak<1>(1, "1 wave/SIMD");
    peak<2>(2, "2 waves/SIMD");
    peak<3>(3, "3 waves/SIMD");
    peak<4>(4, "4 waves/SIMD");
    return 0;
  }
  if (getenv("LAB")) {
    for (int rep = 0; rep < 2; ++rep) {
      timeit(OP_NN, 8192, 8192, 8192, TRI_NONE, 0, 1, "square NN 8192");
      timeit(OP_NN, 32768, 2048, 2048, TRI_KHI_BN, 0, 1, "K*Uinv triu");
      timeit(OP_NT, 32768, 2048, 2048, TRI_KLO_BN, 0, 1, "V*Uinv^T triu");
      timeit(OP_TN, 2048, 2048, 1000064, TRI_NONE, 1, 256, "syrk upper K=1M ks256 scaled", true);
      timeit(OP_NN, 16384, 16384, 8192, TRI_NONE, 0, 1, "NN 16384x16384x8192");
      timeit(OP_NN, 32768, 2048, 1088, TRI_NONE, 0, 1, "full, K=1088 (68 stages)");
      timeit(OP_NN, 32768, 2048, 128, TRI_NONE, 0, 1, "full, K=128 (8 stages)");
      timeit(OP_NN, 32768, 2048, 512, TRI_NONE, 0, 1, "full, K=512 (32 stages)");
      timeit(OP_NN, 32768, 2048, 2048, TRI_NONE, 0, 1, "full, K=2048 (128 stages)");
    }
    return 0;
  }
  if (getenv("F32")) {
    int bad = 0;
    bad += check_f32(OP_NN, 256, 256, 64, TRI_NONE, false, false);
    bad += check_f32(OP_NT, 256, 384, 96, TRI_NONE, false, false);
    bad += check_f32(OP_TN, 256, 256, 160, TRI_NONE, true, false);
    bad += check_f32(OP_NN, 384, 256, 256, TRI_KHI_BN, false, false);
    bad += check_f32(OP_NT, 384, 256, 256, TRI_KLO_BN, false, true);
    bad += check_f32(OP_TN, 128, 128, 32, TRI_NONE, false, false);
    bad += check_f32(OP_NN, 8192, 512, 512, TRI_KHI_BN, false, false);
    bad += check_f32(OP_NT, 8192, 512, 512, TRI_KLO_BN, false, true);
    bad += check_syrk_cs<float>(256, 1056, 8);
    bad += check_syrk_cs<float>(384, 4096 + 96, 3);
    bad += check_two_phase<float>(384, 256, 256);
    bad += check_two_phase<float>(8192, 512, 512);
    printf("f32 checks failed: %d\n", bad);
    timeit_f32(OP_NN, 8192, 8192, 8192, TRI_NONE, 0, 1, "square NN 8192");
    timeit_f32(OP_NN, 32768, 4096, 4096, TRI_KHI_BN, 0, 1, "K*Uinv triu m4096");
    timeit_f32(OP_NT, 32768, 4096, 4096, TRI_KLO_BN, 0, 1, "V*Uinv^T triu m4096");
    timeit_f32(OP_TN, 4096, 4096, 262144, TRI_NONE, 1, 64, "syrk m4096 K=256k ks64");
    return bad;
  }
  int bad = 0;
  bad += check(OP_NN, 256, 256, 64, TRI_NONE, false);
  bad += check(OP_NT, 256, 384, 48, TRI_NONE, false);
  bad += check(OP_TN, 256, 256, 80, TRI_NONE, true);
  bad += check(OP_NN, 384, 256, 256, TRI_KHI_BN, false);
  bad += check(OP_NT, 384, 256, 256, TRI_KLO_BN, false);
  bad += check(OP_NN, 256, 128, 256, TRI_KLO_BM, false);
  bad += check(OP_NT, 256, 256, 256, TRI_KLO_MAX, false);
  bad += check(OP_TN, 128, 128, 16, TRI_NONE, false);
  bad += check(OP_NN, 128, 256, 48, TRI_NONE, false);
  bad += check(OP_NN, 8192, 512, 512, TRI_KHI_BN, false);   // 64 row panels: the paired order (ORD=3) applies
  bad += check(OP_NT, 8192, 512, 512, TRI_KLO_BN, false);
  bad += check(OP_NN, 32768, 1024, 1024, TRI_KHI_BN, false);  // 1024 pairs (ORD=3): several residency rounds
  bad += check(OP_NT, 32768, 1024, 1024, TRI_KLO_BN, false);
  if (getenv("RP")) {
    bad += check(OP_NN, 8192, 2048, 2048, TRI_KHI_BN, false);
    bad += check(OP_NT, 8192, 2048, 2048, TRI_KLO_BN, false);
    printf("checks failed: %d\n", bad);
    return bad;
  }
  bad += check_syrk_cs<double>(256, 1040, 8);
  bad += check_syrk_cs<double>(384, 4096 + 48, 3);
  bad += check_two_phase<double>(384, 256, 256);
  bad += check_two_phase<double>(8192, 512, 512);    // the paired order
  bad += check_two_phase<double>(32768, 1024, 1024);
  printf("checks failed: %d\n", bad);
  timeit(OP_NN, 8192, 8192, 8192, TRI_NONE, 0, 1, "square NN 8192");
  timeit(OP_NN, 32768, 2048, 2048, TRI_NONE, 0, 1, "K*Uinv full");
  timeit(OP_NN, 32768, 2048, 2048, TRI_KHI_BN, 0, 1, "K*Uinv triu");
  timeit(OP_NT, 32768, 2048, 2048, TRI_KLO_BN, 0, 1, "V*Uinv^T triu");
  timeit(OP_TN, 2048, 2048, 32768, TRI_NONE, 1, 8, "syrk upper ks8");
  timeit(OP_TN, 2048, 2048, 262144, TRI_NONE, 1, 64, "syrk upper K=256k ks64");
  if (getenv("BIG")) {
    timeit(OP_TN, 2048, 2048, 1000064, TRI_NONE, 1, 64, "syrk upper K=1M ks64");
    timeit(OP_TN, 2048, 2048, 1000064, TRI_NONE, 1, 64, "syrk upper K=1M ks64 scaled", true);
    timeit(OP_TN, 2048, 2048, 1000064, TRI_NONE, 1, 248, "syrk upper K=1M ks248");
    timeit(OP_TN, 2048, 2048, 524288, TRI_NONE, 1, 64, "syrk upper K=512k ks64");
    timeit(OP_TN, 2048, 2048, 524288, TRI_NONE, 1, 128, "syrk upper K=512k ks128");
    timeit(OP_TN, 2048, 2048, 131072, TRI_NONE, 1, 32, "syrk upper K=128k ks32");
    timeit(OP_TN, 2048, 2048, 131072, TRI_NONE, 1, 64, "syrk upper K=128k ks64");
    return 0;
  }
  timeit(OP_TN, 2048, 2048, 32768, TRI_NONE, 1, 16, "syrk upper ks16");
  timeit(OP_NN, 8192, 2048, 2048, TRI_KHI_BN, 0, 1, "K*Uinv triu small chunk");
  timeit(OP_NN, 131072, 2048, 2048, TRI_KHI_BN, 0, 1, "K*Uinv triu big chunk");
  timeit(OP_NN, 32768, 4096, 4096, TRI_KHI_BN, 0, 1, "K*Uinv triu m4096");
  return bad;
}
