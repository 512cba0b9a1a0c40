// Timing/ablation harness for the diagonal-block Cholesky kernel (development aid).
#include <vector>
#include <cmath>
#include <cstdlib>
#include "../gpr_amd/csrc/kernels.h"
using namespace gprhip;
namespace gprhip { void launch_potrf_diag_flags(double* A, int mp, int j, double* dinv, int* info, int flags, hipStream_t s); }
int main() {
  const int n = 128;
  std::vector<double> G(n * n), A(n * n);
  for (auto& v : G) v = (double)rand() / RAND_MAX - 0.5;
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = 0; for (int k = 0; k < n; ++k) s += G[i*n+k]*G[j*n+k]; A[i*n+j] = s + (i==j ? 1.0 : 0.0); }
  double *dA, *dD; int* dI;
  hipMalloc(&dA, n*n*8); hipMalloc(&dD, n*n*8); hipMalloc(&dI, 8); hipMemset(dI, 0, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int flags : {0, 1, 2, 3, 2 + 4, 2 + 8, 2 + 16, 2 + 4 + 8 + 16}) {
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      hipMemcpy(dA, A.data(), n*n*8, hipMemcpyHostToDevice);
      hipEventRecord(e0, 0);
      launch_potrf_diag_flags(dA, n, 0, dD, dI, flags, 0);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
    }
    printf("flags=%d (1 skip factor, 2 skip invert, 4 skip diag16, 8 skip panel solve, 16 skip trailing): %.1f us\n", flags, best * 1e3);
  }
  // correctness of the full kernel
  hipMemcpy(dA, A.data(), n*n*8, hipMemcpyHostToDevice);
  launch_potrf_diag_flags(dA, n, 0, dD, dI, 0, 0);
  std::vector<double> U(n*n), D(n*n);
  hipMemcpy(U.data(), dA, n*n*8, hipMemcpyDeviceToHost); hipMemcpy(D.data(), dD, n*n*8, hipMemcpyDeviceToHost);
  double e1m = 0, e2m = 0;
  for (int i = 0; i < n; ++i) for (int j = i; j < n; ++j) { double s = 0; for (int k = 0; k <= i; ++k) s += U[k*n+i]*U[k*n+j]; e1m = std::max(e1m, fabs(s - A[i*n+j])); }
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = 0; for (int k = 0; k < n; ++k) s += U[i*n+k]*D[k*n+j]; e2m = std::max(e2m, fabs(s - (i==j))); }
  printf("max |U^T U - A| = %.2e   max |U Dinv - I| = %.2e\n", e1m, e2m);
  return 0;
}
