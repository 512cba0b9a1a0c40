// Small kernels of the posterior paths either side of the evidence evaluation:
// training-set statistics, predictive covariance matrices, covariance samplers.
//   Stats               lib/fitc_gp.ml:304-374
//   FITC_/FIC_covariances  lib/fitc_gp.ml:533-627
//   Common_cov_sampler  lib/fitc_gp.ml:656-697
// All of them are HBM-bound elementwise / reduction work; the contractions run on the MFMA engine.
#include "kernels.h"

namespace gprhip {

namespace {

__device__ inline double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ inline double wave_max_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}

// partial[block][4] = { sum (y-mean)^2, sum |y-mean|, max |y-mean|, sum y^2 } over the block's 256 rows
__global__ __launch_bounds__(256) void residual_stats_kernel(const double* __restrict__ y,
                                                             const double* __restrict__ mean, int rows,
                                                             double* __restrict__ partial) {
  __shared__ double red[4][4];
  const int i = blockIdx.x * 256 + threadIdx.x;
  double sse = 0.0, sad = 0.0, mad = 0.0, sy2 = 0.0;
  if (i < rows) {
    const double yi = y[i];
    const double diff = yi - mean[i];
    sse = diff * diff;
    sad = fabs(diff);
    mad = sad;
    sy2 = yi * yi;
  }
  sse = wave_sum_d(sse);
  sad = wave_sum_d(sad);
  mad = wave_max_d(mad);
  sy2 = wave_sum_d(sy2);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) {
    red[wv][0] = sse;
    red[wv][1] = sad;
    red[wv][2] = mad;
    red[wv][3] = sy2;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const int k = threadIdx.x;
    double out;
    if (k == 2) out = fmax(fmax(red[0][k], red[1][k]), fmax(red[2][k], red[3][k]));
    else out = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
    partial[(int64_t)blockIdx.x * 4 + k] = out;
  }
}

// out (np x np, row-major, symmetric) from the upper triangle of a Fortran nt x nt matrix (element (r,c),
// r <= c, at in[c*ld + r]); diagonal += add; padding rows/columns: identity.
__global__ __launch_bounds__(256) void sym_from_upper_kernel(const double* __restrict__ in, int64_t ld, int nt,
                                                             double* __restrict__ out, int np, double add) {
  const int c = blockIdx.x * 16 + (threadIdx.x & 15);
  const int r = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (r >= np || c >= np) return;
  double v;
  if (r < nt && c < nt) {
    v = (r <= c) ? in[(int64_t)c * ld + r] : in[(int64_t)r * ld + c];
    if (r == c) v += add;
  } else {
    v = (r == c) ? 1.0 : 0.0;
  }
  out[(int64_t)r * np + c] = v;
}

// C[i][i] += (vec ? vec[i] : 0) + add   for i < n
__global__ __launch_bounds__(256) void add_diag_kernel(double* __restrict__ C, int64_t ld, int n,
                                                       const double* __restrict__ vec, double add) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) C[(int64_t)i * ld + i] += (vec ? vec[i] : 0.0) + add;
}

// S[s][i] += v[i]   (S row-major [ns][ld]; i < n)
__global__ __launch_bounds__(256) void add_row_vector_kernel(double* __restrict__ S, int64_t ld, int ns, int n,
                                                             const double* __restrict__ v) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int s = blockIdx.y;
  if (i < n && s < ns) S[(int64_t)s * ld + i] += v[i];
}

// out[i] = a - x[i]
__global__ __launch_bounds__(256) void const_minus_kernel(const double* __restrict__ x, int n, double a,
                                                          double* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = a - x[i];
}

}  // namespace

int residual_stat_blocks(int rows) { return (rows + 255) / 256; }

void launch_residual_stats(const double* y, const double* mean, int rows, double* partial, hipStream_t s) {
  hipLaunchKernelGGL(residual_stats_kernel, dim3(residual_stat_blocks(rows)), dim3(256), 0, s, y, mean, rows,
                     partial);
  GPR_HIP(hipGetLastError());
}

void launch_sym_from_upper(const double* in, int64_t ld, int nt, double* out, int np, double add,
                           hipStream_t s) {
  hipLaunchKernelGGL(sym_from_upper_kernel, dim3((np + 15) / 16, (np + 15) / 16), dim3(256), 0, s, in, ld, nt,
                     out, np, add);
  GPR_HIP(hipGetLastError());
}

void launch_add_diag(double* C, int64_t ld, int n, const double* vec, double add, hipStream_t s) {
  hipLaunchKernelGGL(add_diag_kernel, dim3((n + 255) / 256), dim3(256), 0, s, C, ld, n, vec, add);
  GPR_HIP(hipGetLastError());
}

void launch_add_row_vector(double* S, int64_t ld, int ns, int n, const double* v, hipStream_t s) {
  hipLaunchKernelGGL(add_row_vector_kernel, dim3((n + 255) / 256, ns), dim3(256), 0, s, S, ld, ns, n, v);
  GPR_HIP(hipGetLastError());
}

void launch_const_minus(const double* x, int n, double a, double* out, hipStream_t s) {
  hipLaunchKernelGGL(const_minus_kernel, dim3((n + 255) / 256), dim3(256), 0, s, x, n, a, out);
  GPR_HIP(hipGetLastError());
}

}  // namespace gprhip
