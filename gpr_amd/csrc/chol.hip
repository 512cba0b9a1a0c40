// Diagonal-block kernels of the blocked upper Cholesky (dpotrf `U) and triangular inverse,
// plus the small m-vector helpers.  The trailing updates and panel solves run on the MFMA
// engine; these kernels handle the 128 x 128 diagonal blocks in LDS (one workgroup each).
// Reference call sites: lib/fitc_gp.ml:53-57 (potrf of K_m + jitter), lib/utils.ml:95-113.
#include "kernels.h"

namespace gprhip {

constexpr int NB = TILE;        // 128
constexpr int LDT = NB + 1;     // LDS row stride (bank spread)
constexpr int POTRF_LDS = (NB * LDT) * 8 + 16;  // + flag word

// A = U^T U in place on block j; strict lower of the block zeroed; dinv = inv(U_jj).
__global__ __launch_bounds__(256) void potrf_diag_kernel(double* __restrict__ A, int mp, int j,
                                                         double* __restrict__ dinv,
                                                         int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) double T[];  // [NB][LDT]
  const int tid = threadIdx.x;
  double* Ab = A + (int64_t)j * NB * mp + (int64_t)j * NB;
  for (int idx = tid; idx < NB * NB; idx += 256) {
    int r = idx / NB, c = idx % NB;
    T[r * LDT + c] = Ab[(int64_t)r * mp + c];
  }
  __syncthreads();
  int& bad = *reinterpret_cast<int*>(T + NB * LDT);  // keep all LDS in the one dynamic array
  if (tid == 0) bad = 0;
  __syncthreads();
  // right-looking unblocked Cholesky on the upper triangle
  const int c = tid & (NB - 1);   // column owned
  const int rh = tid >> 7;        // 0/1: which half of the rows
  for (int k = 0; k < NB; ++k) {
    double dkk = T[k * LDT + k];
    if (!(dkk > 0.0)) {
      if (tid == 0) bad = k + 1;
      dkk = 1.0;  // keep going with finite numbers; caller reads `info`
    }
    const double piv = sqrt(dkk);
    __syncthreads();
    if (rh == 0 && c >= k) T[k * LDT + c] = (c == k) ? piv : T[k * LDT + c] / piv;
    __syncthreads();
    if (c > k) {
      const double ukc = T[k * LDT + c];
      for (int r = k + 1 + rh; r <= c; r += 2) T[r * LDT + c] -= T[k * LDT + r] * ukc;
    }
    __syncthreads();
  }
  if (tid == 0 && bad != 0) atomicCAS(info, 0, j * NB + bad);
  // write U back (zero strict lower of the block)
  for (int idx = tid; idx < NB * NB; idx += 256) {
    int r = idx / NB, cc = idx % NB;
    Ab[(int64_t)r * mp + cc] = (cc >= r) ? T[r * LDT + cc] : 0.0;
  }
  // inverse of the upper-triangular block: column cc of X solves U x = e_cc by back substitution,
  // one thread per column.  The strict-lower part of T is free (U lives in the upper part), so
  // thread cc keeps its partial solution x_k (k < cc) in T[cc][k].
  __syncthreads();
  if (tid < NB) {
    const int cc = tid;
    double* X = dinv + cc;  // column cc, row stride NB
    const double xcc = 1.0 / T[cc * LDT + cc];
    for (int r = cc - 1; r >= 0; --r) {
      double s = T[r * LDT + cc] * xcc;  // k = cc term
      for (int k = r + 1; k < cc; ++k) s += T[r * LDT + k] * T[cc * LDT + k];  // scratch holds x_k
      T[cc * LDT + r] = -s / T[r * LDT + r];
    }
    for (int r = 0; r < NB; ++r) {
      double v = 0.0;
      if (r == cc) v = xcc;
      else if (r < cc) v = T[cc * LDT + r];
      X[(int64_t)r * NB] = v;
    }
  }
}

__global__ void zero_strict_lower_kernel(double* __restrict__ A, int mp) {
  int c = blockIdx.x * 256 + threadIdx.x;
  int r = blockIdx.y;
  if (c < mp && c < r) A[(int64_t)r * mp + c] = 0.0;
}

__global__ void copy_block_kernel(const double* __restrict__ src, int64_t lds, double* __restrict__ dst,
                                  int64_t ldd, int rows, int cols) {
  int c = blockIdx.x * 256 + threadIdx.x;
  int r = blockIdx.y;
  if (c < cols && r < rows) dst[(int64_t)r * ldd + c] = src[(int64_t)r * lds + c];
}

__global__ __launch_bounds__(256) void logdet_kernel(const double* __restrict__ A, int mp, int m,
                                                     double* __restrict__ out) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < m; i += 256) s += log(A[(int64_t)i * mp + i]);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0] + red[0];
}

// One block per 64 output entries; a wavefront-wide dot per entry would be overkill for m <= 8k.
__global__ __launch_bounds__(256) void triu_matvec_kernel(const double* __restrict__ A, int mp,
                                                          const double* __restrict__ x,
                                                          double* __restrict__ y, int trans) {
  // one wavefront per output element
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= mp) return;
  double s = 0.0;
  if (!trans) {
    for (int k = i + lane; k < mp; k += 64) s += A[(int64_t)i * mp + k] * x[k];
  } else {
    for (int k = lane; k <= i; k += 64) s += A[(int64_t)k * mp + i] * x[k];
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) y[i] = s;
}

void launch_potrf_diag(double* A, int mp, int j, double* dinv, int* info, hipStream_t s) {
  static bool attr = false;
  if (!attr) {
    GPR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&potrf_diag_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, POTRF_LDS));
    attr = true;
  }
  hipLaunchKernelGGL(potrf_diag_kernel, dim3(1), dim3(256), POTRF_LDS, s, A, mp, j, dinv, info);
  GPR_HIP(hipGetLastError());
}

void launch_zero_strict_lower(double* A, int mp, hipStream_t s) {
  hipLaunchKernelGGL(zero_strict_lower_kernel, dim3((mp + 255) / 256, mp), dim3(256), 0, s, A, mp);
  GPR_HIP(hipGetLastError());
}

void launch_copy_block(const double* src, int64_t lds, double* dst, int64_t ldd, int rows, int cols,
                       hipStream_t s) {
  hipLaunchKernelGGL(copy_block_kernel, dim3((cols + 255) / 256, rows), dim3(256), 0, s, src, lds, dst,
                     ldd, rows, cols);
  GPR_HIP(hipGetLastError());
}

void launch_logdet(const double* A, int mp, int m, double* out, hipStream_t s) {
  hipLaunchKernelGGL(logdet_kernel, dim3(1), dim3(256), 0, s, A, mp, m, out);
  GPR_HIP(hipGetLastError());
}

void launch_triu_matvec(const double* A, int mp, const double* x, double* y, int trans, hipStream_t s) {
  hipLaunchKernelGGL(triu_matvec_kernel, dim3((mp + 3) / 4), dim3(256), 0, s, A, mp, x, y, trans);
  GPR_HIP(hipGetLastError());
}

}  // namespace gprhip
