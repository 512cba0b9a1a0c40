// Timing/ablation harness for the diagonal-block Cholesky kernel (development aid).
#include <vector>
#include <cmath>
#include <cstdlib>
#include "../gpr_amd/csrc/kernels.h"
using namespace gprhip;
int main() {
  const int n = 128;
  std::vector<double> G(n * n), A(n * n);
  for (auto& v : G) v = (double)rand() / RAND_MAX - 0.5;
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = 0; for (int k = 0; k < n; ++k) s += G[i*n+k]*G[j*n+k]; A[i*n+j] = s + (i==j ? 1.0 : 0.0); }
  // M_REAL=<m>: the block as a problem with m < 128 inducing points presents it (identity padding), micro-panels trimmed
  const int m_real = getenv("M_REAL") ? atoi(getenv("M_REAL")) : 0;
  if (m_real > 0)
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j)
        if (i >= m_real || j >= m_real) A[i*n+j] = (i == j) ? 1.0 : 0.0;
  double *dA, *dD; int* dI;
  hipMalloc(&dA, n*n*8); hipMalloc(&dD, n*n*8); hipMalloc(&dI, 8); hipMemset(dI, 0, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int flags : {0, 1, 2, 3, 2 + 32}) {
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      hipMemcpy(dA, A.data(), n*n*8, hipMemcpyHostToDevice);
      hipEventRecord(e0, 0);
      launch_potrf_diag_flags(dA, n, 0, dD, dI, flags, 0, m_real);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
    }
    printf("flags=%d (1 skip factor, 2 skip invert, 32 factor-only step of the blocked factorisation): %.1f us\n", flags, best * 1e3);
  }
  if (getenv("TS")) {  // phase stamps of the factor (s_memtime ticks, thread 0): chain | barrier | panel | first-row update
    for (int fl : {2 + 256, 256}) {
      hipMemcpy(dA, A.data(), n*n*8, hipMemcpyHostToDevice);
      launch_potrf_diag_flags(dA, n, 0, dD, dI, fl, 0, m_real);
      hipDeviceSynchronize();
      unsigned long long t[64];
      potrf_fetch_timestamps(t);
      printf("flags=%d  ticks of wavefront 0 per micro-panel [chain + stores | barrier 1 | X tile | barrier 2 + diagonal tile + relayout]:\n", fl);
      for (int k = 0; k < 8 && t[4 + 4 * k] > t[0]; ++k)
        printf("  k=%d: %5llu %5llu %5llu %5llu\n", k, t[1 + 4 * k] - t[4 * k], t[2 + 4 * k] - t[1 + 4 * k], t[3 + 4 * k] - t[2 + 4 * k],
               t[4 + 4 * k] - t[3 + 4 * k]);
    }
  }
  // correctness of the full kernel
  hipMemcpy(dA, A.data(), n*n*8, hipMemcpyHostToDevice);
  launch_potrf_diag_flags(dA, n, 0, dD, dI, 0, 0, m_real);
  std::vector<double> U(n*n), D(n*n);
  hipMemcpy(U.data(), dA, n*n*8, hipMemcpyDeviceToHost); hipMemcpy(D.data(), dD, n*n*8, hipMemcpyDeviceToHost);
  double e1m = 0, e2m = 0;
  for (int i = 0; i < n; ++i) for (int j = i; j < n; ++j) { double s = 0; for (int k = 0; k <= i; ++k) s += U[k*n+i]*U[k*n+j]; e1m = std::max(e1m, fabs(s - A[i*n+j])); }
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = 0; for (int k = 0; k < n; ++k) s += U[i*n+k]*D[k*n+j]; e2m = std::max(e2m, fabs(s - (i==j))); }
  printf("max |U^T U - A| = %.2e   max |U Dinv - I| = %.2e\n", e1m, e2m);

  // the whole blocked factorisation (potrf_upper_blocked: factor-only diagonal kernel, substitution panel, small-tile
  // trailing update, block inverses at the end) against its definition, and its time
  for (int m : {128, 256, 384, 512, 1024, 2048, 4096}) {
    const int nb = m / 128;
    std::vector<double> B((size_t)m * 64), S((size_t)m * m);
    for (auto& v : B) v = (double)rand() / RAND_MAX - 0.5;
    for (int i = 0; i < m; ++i)
      for (int j = i; j < m; ++j) {
        double s = 0;
        for (int k = 0; k < 64; ++k) s += B[(size_t)i * 64 + k] * B[(size_t)j * 64 + k];
        S[(size_t)i * m + j] = s + (i == j ? 1e-2 * (1 + i % 7) : 0.0);
        if (j > i) S[(size_t)j * m + i] = 1e300;  // the strict lower triangle must never be read
      }
    double *dS, *dV; int* dJ;
    hipMalloc(&dS, (size_t)m * m * 8); hipMalloc(&dV, (size_t)nb * 128 * 128 * 8); hipMalloc(&dJ, 8); hipMemset(dJ, 0, 8);
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
      hipMemcpy(dS, S.data(), (size_t)m * m * 8, hipMemcpyHostToDevice);
      hipEventRecord(e0, 0);
      potrf_upper_blocked(0, dS, m, dV, dJ);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
    }
    int hinfo = -1;
    hipMemcpy(&hinfo, dJ, 4, hipMemcpyDeviceToHost);
    std::vector<double> Um((size_t)m * m), Dv((size_t)nb * 128 * 128);
    hipMemcpy(Um.data(), dS, (size_t)m * m * 8, hipMemcpyDeviceToHost);
    hipMemcpy(Dv.data(), dV, Dv.size() * 8, hipMemcpyDeviceToHost);
    double ea = 0, ed = 0, amax = 0;
    if (m <= 2048) {
      // U^T U against A on the upper triangle: rows of U^T U from the transposed factor (contiguous inner loops)
      std::vector<double> Ut((size_t)m * m, 0.0);
      for (int k = 0; k < m; ++k) for (int j = k; j < m; ++j) Ut[(size_t)j * m + k] = Um[(size_t)k * m + j];
      for (int i = 0; i < m; ++i)
        for (int j = i; j < m; ++j) {
          double s = 0;
          const double *a = &Ut[(size_t)i * m], *b = &Ut[(size_t)j * m];
          for (int k = 0; k <= i; ++k) s += a[k] * b[k];
          ea = std::max(ea, fabs(s - S[(size_t)i * m + j]));
          amax = std::max(amax, fabs(S[(size_t)i * m + j]));
        }
    }
    for (int jb = 0; jb < nb; ++jb)
      for (int i = 0; i < 128; ++i)
        for (int j = 0; j < 128; ++j) {
          double s = 0;
          for (int k = i; k < 128; ++k) s += Um[(size_t)(jb * 128 + i) * m + jb * 128 + k] * Dv[(size_t)jb * 16384 + k * 128 + j];
          ed = std::max(ed, fabs(s - (i == j)));
        }
    printf("blocked potrf m=%d: %.1f us  info=%d  max |U^T U - A| / max|A| = %.2e  max |U_jj Dinv_j - I| = %.2e\n", m,
           best * 1e3, hinfo, amax > 0 ? ea / amax : 0.0, ed);
    // factor + inverse in one pass (identity right-hand side carried along): U X = I, X upper triangular
    double *dY, *dX;
    hipMalloc(&dY, (size_t)m * m * 8); hipMalloc(&dX, (size_t)m * m * 8);
    hipMemset(dX, 0xff, (size_t)m * m * 8);
    float besti = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
      hipMemcpy(dS, S.data(), (size_t)m * m * 8, hipMemcpyHostToDevice);
      hipEventRecord(e0, 0);
      potrf_upper_blocked(0, dS, m, dV, dJ, dY, dX);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); besti = std::min(besti, ms);
    }
    {  // the same with the look-ahead split of the trailing update (side stream): timing, and the results below are its
      PotrfAux aux;
      aux.min_rest = getenv("LOOKAHEAD") ? atoi(getenv("LOOKAHEAD")) : 24;
      hipStream_t ms;  // (not the null stream: it synchronises with every other stream)
      hipStreamCreateWithFlags(&ms, hipStreamNonBlocking);
      hipStreamCreateWithFlags(&aux.side, hipStreamNonBlocking);
      for (int k = 0; k < 2; ++k) {
        hipEventCreateWithFlags(&aux.ev_panel[k], hipEventDisableTiming);
        hipEventCreateWithFlags(&aux.ev_rest[k], hipEventDisableTiming);
      }
      float bestl = 1e9, best1 = 1e9;
      for (int rep = 0; rep < 8; ++rep) {
        hipMemcpy(dS, S.data(), (size_t)m * m * 8, hipMemcpyHostToDevice);
        hipMemset(dX, 0xff, (size_t)m * m * 8);
        hipDeviceSynchronize();
        hipEventRecord(e0, ms);
        potrf_upper_blocked(ms, dS, m, dV, dJ, dY, dX, 0, (rep & 1) ? &aux : nullptr);
        hipEventRecord(e1, ms); hipEventSynchronize(e1);
        float t; hipEventElapsedTime(&t, e0, e1);
        if (rep & 1) bestl = std::min(bestl, t); else best1 = std::min(best1, t);
      }
      printf("  with look-ahead (side stream) m=%d: %.1f us (one stream %.1f)\n", m, bestl * 1e3, best1 * 1e3);
      hipStreamSynchronize(aux.side);
      hipStreamDestroy(aux.side);
      hipStreamDestroy(ms);
    }
    std::vector<double> U2((size_t)m * m), Xi((size_t)m * m);
    hipMemcpy(U2.data(), dS, (size_t)m * m * 8, hipMemcpyDeviceToHost);
    hipMemcpy(Xi.data(), dX, (size_t)m * m * 8, hipMemcpyDeviceToHost);
    double eu = 0, ei = 0, el = 0;
    for (int i = 0; i < m; ++i) for (int j = i; j < m; ++j) eu = std::max(eu, fabs(U2[(size_t)i * m + j] - Um[(size_t)i * m + j]));
    for (int i = 0; i < m; ++i) for (int j = 0; j < i; ++j) el = std::max(el, fabs(Xi[(size_t)i * m + j]));
    if (m <= 2048) {
      std::vector<double> Xt((size_t)m * m);  // columns of X contiguous
      for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) Xt[(size_t)j * m + i] = Xi[(size_t)i * m + j];
      for (int i = 0; i < m; ++i)
        for (int j = i; j < m; ++j) {
          double s = 0;
          const double *u = &U2[(size_t)i * m], *x = &Xt[(size_t)j * m];
          for (int k = i; k <= j; ++k) s += u[k] * x[k];
          ei = std::max(ei, fabs(s - (i == j)));
        }
    }
    printf("blocked potrf+inverse m=%d: %.1f us  factor differs by %.1e  max |U X - I| = %.2e  max |strict lower of X| = %.1e\n",
           m, besti * 1e3, eu, ei, el);
    // the same as ONE persistent launch with device-side dependencies (potrf_upper_chain): bit-identical factor and inverse
    if (m >= 256 && m <= 8192) {
      PotrfChain* ch = potrf_chain_create(m);
      double* dX2;
      hipMalloc(&dX2, (size_t)m * m * 8);
      hipMemset(dX2, 0xff, (size_t)m * m * 8);
      float bestc = 1e9;
      const int reps = getenv("CHAIN_REPS") ? atoi(getenv("CHAIN_REPS")) : 6;
      for (int rep = 0; rep < reps; ++rep) {
        hipMemcpy(dS, S.data(), (size_t)m * m * 8, hipMemcpyHostToDevice);
        hipMemset(dJ, 0, 8);
        hipEventRecord(e0, 0);
        potrf_upper_chain(0, ch, dS, m, dV, dJ, dY, dX2);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); bestc = std::min(bestc, ms);
      }
      int cinfo = -1;
      hipMemcpy(&cinfo, dJ, 4, hipMemcpyDeviceToHost);
      std::vector<double> U3((size_t)m * m), X3((size_t)m * m);
      hipMemcpy(U3.data(), dS, (size_t)m * m * 8, hipMemcpyDeviceToHost);
      hipMemcpy(X3.data(), dX2, (size_t)m * m * 8, hipMemcpyDeviceToHost);
      double du = 0, dx = 0;
      for (int i = 0; i < m; ++i)
        for (int j = 0; j < m; ++j) {
          if (j >= i) du = std::max(du, fabs(U3[(size_t)i * m + j] - U2[(size_t)i * m + j]));
          dx = std::max(dx, fabs(X3[(size_t)i * m + j] - Xi[(size_t)i * m + j]));
        }
      printf("chain potrf+inverse m=%d: %.1f us (stepwise %.1f)  info=%d%s  factor differs by %.1e  inverse differs by %.1e\n", m,
             bestc * 1e3, besti * 1e3, cinfo, cinfo == POTRF_CHAIN_ABORT_CODE ? " (ABORTED)" : "", du, dx);
      if (getenv("TRACE") && atoi(getenv("TRACE")) == m) {  // the launch's timeline (100 MHz wall clock, us from the first stamp)
        const int nt = potrf_chain_tasks(ch, nullptr);
        std::vector<int> tk((size_t)nt * 4);
        potrf_chain_tasks(ch, tk.data());
        unsigned long long* dT;
        hipMalloc(&dT, (size_t)(nt + nb) * 32);
        hipMemset(dT, 0, (size_t)(nt + nb) * 32);
        hipMemcpy(dS, S.data(), (size_t)m * m * 8, hipMemcpyHostToDevice);
        hipMemset(dJ, 0, 8);
        potrf_upper_chain(0, ch, dS, m, dV, dJ, dY, dX2, 0, dT);
        hipDeviceSynchronize();
        std::vector<unsigned long long> tt((size_t)(nt + nb) * 4);
        hipMemcpy(tt.data(), dT, tt.size() * 8, hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull;
        for (auto v : tt) if (v && v < t0) t0 = v;
        auto us = [&](unsigned long long v) { return v ? (double)(v - t0) * 0.01 : -1.0; };
        const char* names[] = {"P ", "PY", "U ", "UY"};
        const int maxj = getenv("TRACE_STEPS") ? atoi(getenv("TRACE_STEPS")) : 3;
        for (int j = 0; j < nb; ++j) {
          const unsigned long long* d = &tt[(size_t)(nt + j) * 4];
          printf("D(%d)            start %8.2f  deps %8.2f  done %8.2f  flag %8.2f\n", j, us(d[0]), us(d[1]), us(d[2]), us(d[3]));
          if (j >= maxj) continue;
          for (int i = 0; i < nt; ++i)
            if (tk[4 * i + 1] == j && (tk[4 * i] != 2 && tk[4 * i] != 3 || (tk[4 * i + 3] >> 16) == 1 || getenv("TRACE_ALL")))
              printf("  %s(%d,%3d,%3d) #%5d start %8.2f  deps %8.2f  done %8.2f  flag %8.2f\n", names[tk[4 * i]], j, tk[4 * i + 2],
                     tk[4 * i + 3] & 0xffff, i, us(tt[4 * (size_t)i]), us(tt[4 * (size_t)i + 1]), us(tt[4 * (size_t)i + 2]), us(tt[4 * (size_t)i + 3]));
        }
        hipFree(dT);
      }
      hipFree(dX2);
      potrf_chain_destroy(ch);
    }
    hipFree(dY); hipFree(dX);
    hipFree(dS); hipFree(dV); hipFree(dJ);
  }
  return 0;
}
