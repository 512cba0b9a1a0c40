// Row-wise (per training point) kernels of the two streaming passes -- all HBM-bound.
// Wavefront reductions give diag(Q_nn)-type quantities; every cross-block sum goes through
// a partial buffer that is reduced in a fixed order (bit-reproducible run to run).
#include "kernels.h"
#include "exp_fast.h"

namespace gprhip {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// 16 bytes of a row per lane: two doubles or four floats, accumulated in fp64 either way
template <typename TS>
struct RowVec;
template <>
struct RowVec<double> {
  static constexpr int N = 2;
  static __device__ __forceinline__ void load(const double* p, double (&v)[2]) {
    const double2 x = *reinterpret_cast<const double2*>(p);
    v[0] = x.x; v[1] = x.y;
  }
};
template <>
struct RowVec<float> {
  static constexpr int N = 4;
  static __device__ __forceinline__ void load(const float* p, double (&v)[4]) {
    const float4 x = *reinterpret_cast<const float4*>(p);
    v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
  }
};

constexpr int ROWS_PER_BLOCK = 256;  // one training point per thread

int pass1_row_blocks(int rows_p) { return (rows_p + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK; }

// block sums of up to four per-thread values, wavefronts combined in a fixed order -> partial[block][4]
__device__ __forceinline__ void block_sums4(double v0, double v1, double v2, double v3, double* out) {
  __shared__ double red[4][4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  v0 = wave_sum(v0);
  v1 = wave_sum(v1);
  v2 = wave_sum(v2);
  v3 = wave_sum(v3);
  if (lane == 0) {
    red[wv][0] = v0;
    red[wv][1] = v1;
    red[wv][2] = v2;
    red[wv][3] = v3;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const int k = threadIdx.x;
    out[k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
  }
}

// r = k_diag - rowsum(V.^2)            lib/fitc_gp.ml:222-223 (Mat.syrk_diag)
// s = r + sigma2, is = 1/s, sum log s   lib/fitc_gp.ml:155-166
// rowsum(V.^2) arrives as `npart` partial sums per row from the epilogue of the V product, part-major
// ([npart][ld]: adjacent threads read adjacent addresses).
__global__ __launch_bounds__(256) void pass1_rows_kernel(Pass1RowArgs a, int rows_p) {
  const int row = blockIdx.x * ROWS_PER_BLOCK + threadIdx.x;
  double p_log = 0.0, p_y2 = 0.0, p_isr = 0.0;
  if (row < rows_p) {
    if (row < a.rows) {
      double r;
      if (a.part) {
        double s2 = 0.0;
        for (int c = 0; c < a.npart; ++c) s2 += a.part[(int64_t)c * a.ld + row];
        r = a.sf2 - s2;
      } else {
        r = a.r[row];  // Model.update_sigma2 (lib/fitc_gp.ml:234-236): r of the previous evaluation is kept
      }
      const double s = r + a.sigma2;
      const double is = 1.0 / s;
      const double y = a.y ? a.y[row] : 0.0;
      a.r[row] = r;
      a.is[row] = is;
      a.yis[row] = is * y;
      p_log = log(s);
      p_y2 = is * y * y;
      p_isr = is * r;
    } else {
      a.r[row] = 0.0;
      a.is[row] = 0.0;
      a.yis[row] = 0.0;
    }
  }
  block_sums4(p_log, p_y2, p_isr, 0.0, a.partial + (int64_t)blockIdx.x * 4);
}

void launch_pass1_rows(const Pass1RowArgs& a, hipStream_t s) {
  const int rows_p = (int)round_up(a.rows, TILE);
  hipLaunchKernelGGL(pass1_rows_kernel, dim3(pass1_row_blocks(rows_p)), dim3(256), 0, s, a, rows_p);
  GPR_HIP(hipGetLastError());
}

// q_diag, u, w, v of lib/fitc_gp.ml:1048, :1092-1108, :1158-1181 from Q' = K R^-1 (so that
// Q_n = diag(sqrt is) Q'):  q_diag = is*|Q'_i|^2,  u/sqrt(is) = y - Q' b  (b = Q_n^T y~),
// w = is*(y - Q' b),  v1 = is*(1-q) [variational: is*(2 - is*r - q)],  v = v1 - w^2.
// |Q'_i|^2 and Q'_i . b arrive as partial sums from the epilogue of the Q' product (part-major, as above).
__global__ __launch_bounds__(256) void pass2_rows_kernel(Pass2RowArgs a, int rows_p) {
  const int row = blockIdx.x * ROWS_PER_BLOCK + threadIdx.x;
  double p_v = 0.0, p_is = 0.0, p_res = 0.0, p_v1 = 0.0;
  if (row < rows_p) {
    if (row < a.rows) {
      double s2 = 0.0, sb = 0.0;
      for (int c = 0; c < a.npart; ++c) {
        s2 += a.part_sq[(int64_t)c * a.ld + row];
        sb += a.part_dot[(int64_t)c * a.ld + row];
      }
      const double is = a.is[row];
      const double qd = is * s2;
      const double y = a.y ? a.y[row] : 0.0;
      const double res = a.y ? (y - sb) : 0.0;
      const double w = is * res;
      const double v1 = a.variational ? is * (2.0 - is * a.r[row] - qd) : is * (1.0 - qd);
      const double v = v1 - w * w;
      a.w[row] = w;
      a.v[row] = v;
      if (a.es) a.es[row] = qd - v * (a.sf2 - a.r[row]) - w * sb;
      p_v = v;
      p_is = is;
      p_res = w * res;
      p_v1 = v1;
    } else {
      a.w[row] = 0.0;
      a.v[row] = 0.0;
      if (a.es) a.es[row] = 0.0;
    }
  }
  block_sums4(p_v, p_is, p_res, p_v1, a.partial + (int64_t)blockIdx.x * 4);
}

void launch_pass2_rows(const Pass2RowArgs& a, hipStream_t s) {
  const int rows_p = (int)round_up(a.rows, TILE);
  hipLaunchKernelGGL(pass2_rows_kernel, dim3(pass1_row_blocks(rows_p)), dim3(256), 0, s, a, rows_p);
  GPR_HIP(hipGetLastError());
}

template <typename TS>
__global__ __launch_bounds__(256) void row_sumsq_dot_kernel(const TS* __restrict__ M,
                                                            const double* __restrict__ b, int rows, int mp,
                                                            double* __restrict__ sumsq,
                                                            double* __restrict__ dot) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wv;
  if (row >= rows) return;
  const TS* mr = M + (int64_t)row * mp;
  constexpr int NV = RowVec<TS>::N;
  double s2 = 0.0, sb = 0.0;
  for (int c = lane * NV; c < mp; c += 64 * NV) {
    double x[NV];
    RowVec<TS>::load(mr + c, x);
#pragma unroll
    for (int e = 0; e < NV; ++e) {
      s2 += x[e] * x[e];
      if (b) sb += x[e] * b[c + e];
    }
  }
  s2 = wave_sum(s2);
  sb = wave_sum(sb);
  if (lane == 0) {
    if (sumsq) sumsq[row] = s2;
    if (dot) dot[row] = sb;
  }
}

template <typename TS>
void launch_row_sumsq_dot(const TS* M, const double* b, int rows, int mp, double* sumsq, double* dot,
                          hipStream_t s) {
  hipLaunchKernelGGL(row_sumsq_dot_kernel<TS>, dim3((rows + 3) / 4), dim3(256), 0, s, M, b, rows, mp, sumsq,
                     dot);
  GPR_HIP(hipGetLastError());
}
template void launch_row_sumsq_dot<double>(const double*, const double*, int, int, double*, double*, hipStream_t);
template void launch_row_sumsq_dot<float>(const float*, const double*, int, int, double*, double*, hipStream_t);

__global__ void variance_combine_kernel(const double* __restrict__ k, const double* __restrict__ b, int rows,
                                        double sf2, double add, double* __restrict__ var) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r < rows) var[r] = (sf2 - (k[r] - b[r])) + add;  // prior_variance -. (k -. b), lib/fitc_gp.ml:475
}

void launch_variance_combine(const double* k, const double* b, int rows, double sf2, double add,
                             double* var, hipStream_t s) {
  hipLaunchKernelGGL(variance_combine_kernel, dim3((rows + 255) / 256), dim3(256), 0, s, k, b, rows, sf2, add,
                     var);
  GPR_HIP(hipGetLastError());
}

// out[col] (+)= sum_slab partial[slab][col].  A block covers COLS columns with 256/COLS slab lanes
// per column; each lane strides over the slabs, then the lanes are combined in a fixed order.
template <int COLS>
__global__ __launch_bounds__(256) void reduce_rows_kernel(const double* __restrict__ partial, int nslabs,
                                                          int width, double* __restrict__ out,
                                                          int accumulate) {
  constexpr int LANES = 256 / COLS;
  __shared__ double red[LANES][COLS];
  const int tx = threadIdx.x % COLS, ty = threadIdx.x / COLS;
  const int col = blockIdx.x * COLS + tx;
  double acc = 0.0;
  if (col < width) {
#pragma unroll 8
    for (int sl = ty; sl < nslabs; sl += LANES) acc += partial[(int64_t)sl * width + col];
  }
  red[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && col < width) {
    double tot = accumulate ? out[col] : 0.0;
    for (int l = 0; l < LANES; ++l) tot += red[l][tx];
    out[col] = tot;
  }
}

void launch_reduce_rows(const double* partial, int nslabs, int width, double* out, int accumulate,
                        hipStream_t s) {
  if (width <= 8) {
    hipLaunchKernelGGL(reduce_rows_kernel<8>, dim3((width + 7) / 8), dim3(256), 0, s, partial, nslabs,
                       width, out, accumulate);
  } else {
    hipLaunchKernelGGL(reduce_rows_kernel<64>, dim3((width + 63) / 64), dim3(256), 0, s, partial,
                       nslabs, width, out, accumulate);
  }
  GPR_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// Fused gradient pass: never materialises a derivative matrix.  With
//   X   = diag(is) K Binv - diag(v) K Kminv - w t^T          lib/fitc_gp.ml:1204-1206
//   E   = X .* K_nm                                           (K_nm recomputed per element)
//   `Factor 1.  (Log_sf2)       : tr(X^T K)             = sum E                lib/fitc_gp.ml:991
//   `Dense      (Log_ell)       : tr(X^T (K.*D)) ell^-2 = ell^-2 sum E.*D      lib/cov_se_iso.ml:303-314
//   `Sparse_cols (Inducing c,k) : scale * sum_r (x_kr - z_kc) E_rc             lib/cov_se_iso.ml:315-327
// Thread <-> inducing column; rows stream through; per-thread accumulators in registers.
// Rows per slab of the gradient kernels' column partials: 256, or 1024 when a slab carries 32 or more accumulator rows
// (wide points / projection hypers: the partials are then the kernels' main memory traffic).
int grad_slab_rows(int col_rows) { return col_rows >= 32 ? 1024 : 256; }

template <int DT, int DBT, typename TS>
__global__ __launch_bounds__(256) void grad_fused_kernel(GradArgs<TS> a) {
  const ExpK ek = exp_consts();
  __shared__ double red[4][2];
  __shared__ double xs[32][DT];                  // the 32 points being streamed, zero-padded to DT
  __shared__ double xbs[32][DBT > 0 ? DBT : 1];  // their original inputs (Cov_se_fat with tproj)
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int jj = min(j, a.mp - 1);
  const bool live = (j < a.m);
  double z[DT], gx[DT];
  double gb[DBT > 0 ? DBT : 1];
#pragma unroll
  for (int k = 0; k < DT; ++k) {
    z[k] = (k < a.d && live) ? a.Z[(int64_t)jj * a.d + k] : 0.0;
    gx[k] = 0.0;
  }
#pragma unroll
  for (int k = 0; k < DBT; ++k) gb[k] = 0.0;
  double cs = 0.0, sE = 0.0, sED = 0.0;
  const int r0 = blockIdx.y * a.slab;
  const int r1 = min(a.rows, r0 + a.slab);
  for (int rb = r0; rb < r1; rb += 32) {
    __syncthreads();
    for (int idx = threadIdx.x; idx < 32 * DT; idx += 256) {
      const int r = idx / DT, k = idx % DT;
      xs[r][k] = (k < a.d && rb + r < r1) ? a.pts[(int64_t)(rb + r) * a.d + k] : 0.0;
    }
    if (DBT > 0) {
      for (int idx = threadIdx.x; idx < 32 * DBT; idx += 256) {
        const int r = idx / DBT, k = idx % DBT;
        xbs[r][k] = (k < a.D && rb + r < r1) ? a.big[(int64_t)(rb + r) * a.D + k] : 0.0;
      }
    }
    __syncthreads();
    const int nr = min(32, r1 - rb);
    for (int i = 0; i < nr; ++i) {
      const double xv = (double)a.X[(int64_t)(rb + i) * a.mp + jj];
      double dist = 0.0;
#pragma unroll
      for (int k = 0; k < DT; ++k) {
        const double df = xs[i][k] - z[k];
        dist += df * df;
      }
      const double e = live ? xv * exp_fast(a.log_sf2 + a.inv_ell2_05 * dist, ek) : 0.0;
#pragma unroll
      for (int k = 0; k < DT; ++k) gx[k] += xs[i][k] * e;
#pragma unroll
      for (int k = 0; k < DBT; ++k) gb[k] += xbs[i][k] * e;
      cs += e;
      sE += e;
      sED += e * dist;
    }
  }
  if (j < a.mp) {
    double* cp = a.colpart + (int64_t)blockIdx.y * a.col_rows * a.mp;
    cp[j] = cs;
#pragma unroll
    for (int k = 0; k < DT; ++k)
      if (k < a.d) cp[(int64_t)(k + 1) * a.mp + j] = gx[k];
#pragma unroll
    for (int k = 0; k < DBT; ++k)
      if (k < a.D) cp[(int64_t)(a.d + 1 + k) * a.mp + j] = gb[k];
  }
  sE = wave_sum(sE);
  sED = wave_sum(sED);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) {
    red[wv][0] = sE;
    red[wv][1] = sED;
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    const int k = threadIdx.x;
    a.scalpart[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 + k] =
        (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
  }
}

// Multiscale variant of the fused gradient pass (Cov_se_fat with log_multiscales_m05):
//   K_rc = exp(log_sf2 - 1/2 sum_k [(p_kr - z_kc)^2 / ms_kc + log ms_kc])        lib/cov_se_fat.ml:241-251
// extra column accumulators sum_r p_kr^2 E_rc (for `Log_multiscale_m05, :598-622) and, when a projection is
// optimised too, per-row partial sums of E_rc / ms_kc over this wavefront's columns (`Proj with
// multiscales, :585-595), reduced across wavefronts by launch_reduce_rowes.
template <int DT, int DBT, typename TS>
__global__ __launch_bounds__(256) void grad_fused_ms_kernel(GradArgs<TS> a) {
  const ExpK ek = exp_consts();
  __shared__ double red[4];
  __shared__ double xs[32][DT];
  __shared__ double xbs[32][DBT > 0 ? DBT : 1];
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int jj = min(j, a.mp - 1);
  const bool live = (j < a.m);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int nslots = 4 * gridDim.x, slot = 4 * blockIdx.x + wv;
  double z[DT], isc[DT], gx[DT], gxx[DT];
  double gb[DBT > 0 ? DBT : 1];
  double lsum = 0.0;
#pragma unroll
  for (int k = 0; k < DT; ++k) {
    const bool lk = (k < a.d && live);
    z[k] = lk ? a.Z[(int64_t)jj * a.d + k] : 0.0;
    const double sc = lk ? a.ms[(int64_t)jj * a.d + k] : 1.0;
    isc[k] = 1.0 / sc;
    lsum += log(sc);
    gx[k] = 0.0;
    gxx[k] = 0.0;
  }
#pragma unroll
  for (int k = 0; k < DBT; ++k) gb[k] = 0.0;
  double cs = 0.0, sE = 0.0;
  const int r0 = blockIdx.y * a.slab;
  const int r1 = min(a.rows, r0 + a.slab);
  for (int rb = r0; rb < r1; rb += 32) {
    __syncthreads();
    for (int idx = threadIdx.x; idx < 32 * DT; idx += 256) {
      const int r = idx / DT, k = idx % DT;
      xs[r][k] = (k < a.d && rb + r < r1) ? a.pts[(int64_t)(rb + r) * a.d + k] : 0.0;
    }
    if (DBT > 0) {
      for (int idx = threadIdx.x; idx < 32 * DBT; idx += 256) {
        const int r = idx / DBT, k = idx % DBT;
        xbs[r][k] = (k < a.D && rb + r < r1) ? a.big[(int64_t)(rb + r) * a.D + k] : 0.0;
      }
    }
    __syncthreads();
    const int nr = min(32, r1 - rb);
    for (int i = 0; i < nr; ++i) {
      const double xv = (double)a.X[(int64_t)(rb + i) * a.mp + jj];
      double dist = lsum;
#pragma unroll
      for (int k = 0; k < DT; ++k) {
        const double df = xs[i][k] - z[k];
        dist += df * df * isc[k];
      }
      const double e = live ? xv * exp_fast(a.log_sf2 + a.inv_ell2_05 * dist, ek) : 0.0;
#pragma unroll
      for (int k = 0; k < DT; ++k) {
        gx[k] += xs[i][k] * e;
        gxx[k] += xs[i][k] * xs[i][k] * e;
      }
#pragma unroll
      for (int k = 0; k < DBT; ++k) gb[k] += xbs[i][k] * e;
      if (DBT > 0) {
        double* ro = a.rowes + ((int64_t)(rb + i) * nslots + slot) * a.d;
#pragma unroll
        for (int k = 0; k < DT; ++k) {
          if (k < a.d) {
            const double t = wave_sum(e * isc[k]);
            if (lane == 0) ro[k] = t;
          }
        }
      }
      cs += e;
      sE += e;
    }
  }
  if (j < a.mp) {
    double* cp = a.colpart + (int64_t)blockIdx.y * a.col_rows * a.mp;
    cp[j] = cs;
#pragma unroll
    for (int k = 0; k < DT; ++k) {
      if (k < a.d) {
        cp[(int64_t)(k + 1) * a.mp + j] = gx[k];
        cp[(int64_t)(a.d + 1 + a.D + k) * a.mp + j] = gxx[k];
      }
    }
#pragma unroll
    for (int k = 0; k < DBT; ++k)
      if (k < a.D) cp[(int64_t)(a.d + 1 + k) * a.mp + j] = gb[k];
  }
  sE = wave_sum(sE);
  __syncthreads();
  if (lane == 0) red[wv] = sE;
  __syncthreads();
  if (threadIdx.x == 0) {
    double* sp = a.scalpart + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 2;
    sp[0] = (red[0] + red[1]) + (red[2] + red[3]);
    sp[1] = 0.0;
  }
}

__global__ __launch_bounds__(256) void reduce_rowes_kernel(const double* __restrict__ rowes, int rows,
                                                           int nslots, int d, double* __restrict__ es) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)rows * d) return;
  const int64_t r = idx / d;
  const int k = (int)(idx % d);
  double acc = 0.0;
  for (int sl = 0; sl < nslots; ++sl) acc += rowes[(r * nslots + sl) * d + k];
  es[idx] = acc;
}

void launch_reduce_rowes(const double* rowes, int rows, int nslots, int d, double* es, hipStream_t s) {
  hipLaunchKernelGGL(reduce_rowes_kernel, dim3((unsigned)(((int64_t)rows * d + 255) / 256)), dim3(256), 0, s,
                     rowes, rows, nslots, d, es);
  GPR_HIP(hipGetLastError());
}

// ---- point dimensions above 64 ("wide").  K_nm of the chunk is in memory (the caller rebuilds it with the wide
// covariance kernel: the chunk buffer that held X~ is free), so E = X .* K needs no distance loop, and the column
// accumulators sum_r p_kr E_rc run in passes over 32 dimensions (blockIdx.z), first the d kernel-space dimensions,
// then the D input dimensions of a projected kernel.  The squared distance of the Log_ell term is recovered from K:
// |x - z|^2 = (log K - log sf2) / inv_ell2_05.  No multiscales on this path.
template <typename TS>
__global__ __launch_bounds__(256) void grad_wide_kernel(GradArgs<TS> a, const TS* __restrict__ K) {
  __shared__ double red[4][2];
  __shared__ double xs[32][32];
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int jj = min(j, a.mp - 1);
  const bool live = (j < a.m);
  const int nz = (a.d + 31) / 32;                 // passes over kernel-space dimensions
  const bool big = (int)blockIdx.z >= nz;
  const double* src = big ? a.big : a.pts;
  const int ld = big ? a.D : a.d;
  const int dim0 = (big ? (int)blockIdx.z - nz : (int)blockIdx.z) * 32;
  const int nd = min(32, ld - dim0);
  const int out0 = big ? a.d + 1 + dim0 : 1 + dim0;
  double gx[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) gx[k] = 0.0;
  double cs = 0.0, sED = 0.0;
  const int r0 = blockIdx.y * a.slab;
  const int r1 = min(a.rows, r0 + a.slab);
  for (int rb = r0; rb < r1; rb += 32) {
    __syncthreads();
    for (int idx = threadIdx.x; idx < 32 * 32; idx += 256) {
      const int r = idx / 32, k = idx % 32;
      xs[r][k] = (k < nd && rb + r < r1) ? src[(int64_t)(rb + r) * ld + dim0 + k] : 0.0;
    }
    __syncthreads();
    const int nr = min(32, r1 - rb);
    for (int i = 0; i < nr; ++i) {
      const double kv = (double)K[(int64_t)(rb + i) * a.mp + jj];
      const double e = live ? (double)a.X[(int64_t)(rb + i) * a.mp + jj] * kv : 0.0;
#pragma unroll
      for (int k = 0; k < 32; ++k) gx[k] += xs[i][k] * e;
      cs += e;
      if (kv > 0.0) sED += e * ((log(kv) - a.log_sf2) / a.inv_ell2_05);
    }
  }
  if (j < a.mp) {
    double* cp = a.colpart + (int64_t)blockIdx.y * a.col_rows * a.mp;
    if (blockIdx.z == 0) cp[j] = cs;
#pragma unroll
    for (int k = 0; k < 32; ++k)
      if (k < nd) cp[(int64_t)(out0 + k) * a.mp + j] = gx[k];
  }
  if (blockIdx.z != 0) return;
  double sE = wave_sum(cs);
  sED = wave_sum(sED);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) {
    red[wv][0] = sE;
    red[wv][1] = sED;
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    const int k = threadIdx.x;
    a.scalpart[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 + k] =
        (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
  }
}

template <typename TS>
void launch_grad_wide(const GradArgs<TS>& a, const TS* K, hipStream_t s) {
  if (a.ms) {
    set_error("gprhip: Cov_se_fat multiscales support dimensions d, D <= 64");
    throw HipFail{ST_BAD_ARG};
  }
  const int nz = (a.d + 31) / 32 + (a.big ? (a.D + 31) / 32 : 0);
  dim3 grid((a.mp + 255) / 256, (a.rows + a.slab - 1) / a.slab, nz);
  hipLaunchKernelGGL((grad_wide_kernel<TS>), grid, dim3(256), 0, s, a, K);
  GPR_HIP(hipGetLastError());
}
template void launch_grad_wide<double>(const GradArgs<double>&, const double*, hipStream_t);
template void launch_grad_wide<float>(const GradArgs<float>&, const float*, hipStream_t);

// Cov_se_fat `Proj {big; small}` (lib/cov_se_fat.ml:570-596): second term of
//   -tr(X^T dK) = -[ sum_c z_small,c sum_r x_big,r E_rc  -  sum_r x_big,r p_small,r rowsum(E)_r ]
// One workgroup per slab of 256 rows; the slab's rows pass through LDS `rs` at a time (x_big as is, p_small already
// multiplied by its weight), thread t accumulates outputs t, t + 256, ... (eight at a time).
__global__ __launch_bounds__(256) void proj_term2_kernel(const double* __restrict__ X,
                                                         const double* __restrict__ P,
                                                         const double* __restrict__ es, int es_ld, int rows,
                                                         int D, int d, int rs, double* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) double proj_lds[];
  double* const xs = proj_lds;                 // [rs][D]
  double* const pw = proj_lds + (size_t)rs * D;  // [rs][d]
  const int tid = threadIdx.x;
  const int r0 = blockIdx.x * 256, r1 = min(rows, r0 + 256);
  const int nout = D * d;
  for (int o0 = 0; o0 < nout; o0 += 256 * 8) {
    double acc[8];
    int big[8], small[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int idx = min(o0 + q * 256 + tid, nout - 1);
      acc[q] = 0.0;
      big[q] = idx / d;
      small[q] = idx % d;
    }
    for (int rb = r0; rb < r1; rb += rs) {
      const int nr = min(rs, r1 - rb);
      __syncthreads();
      for (int i = tid; i < nr * D; i += 256) xs[i] = X[(int64_t)rb * D + i];
      for (int i = tid; i < nr * d; i += 256) {
        const int r = i / d, k = i % d;
        pw[i] = P[(int64_t)rb * d + i] * es[(int64_t)(rb + r) * es_ld + (es_ld > 1 ? k : 0)];
      }
      __syncthreads();
      for (int r = 0; r < nr; ++r) {
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] += xs[r * D + big[q]] * pw[r * d + small[q]];
      }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int idx = o0 + q * 256 + tid;
      if (idx < nout) part[(int64_t)blockIdx.x * nout + idx] = acc[q];
    }
  }
}

void launch_proj_term2(const double* X, const double* P, const double* es, int es_ld, int rows, int D, int d,
                       double* part, hipStream_t s) {
  int rs = (int)std::min<int64_t>(64, (48 * 1024) / ((int64_t)(D + d) * 8));
  if (rs < 1) {
    set_error("gprhip: Cov_se_fat projection gradient: D + d too large for the staging buffer");
    throw HipFail{ST_BAD_ARG};
  }
  hipLaunchKernelGGL(proj_term2_kernel, dim3((rows + 255) / 256), dim3(256), (size_t)rs * (D + d) * 8, s, X, P, es, es_ld,
                     rows, D, d, rs, part);
  GPR_HIP(hipGetLastError());
}

template <int DT, typename TS>
static void grad_dispatch_big(const GradArgs<TS>& a, dim3 grid, hipStream_t s) {
  if (a.ms) {
    if (!a.big) hipLaunchKernelGGL((grad_fused_ms_kernel<DT, 0, TS>), grid, dim3(256), 0, s, a);
    else if (a.D <= 8) hipLaunchKernelGGL((grad_fused_ms_kernel<DT, 8, TS>), grid, dim3(256), 0, s, a);
    else if (a.D <= 32) hipLaunchKernelGGL((grad_fused_ms_kernel<DT, 32, TS>), grid, dim3(256), 0, s, a);
    else {
      set_error("gprhip: Cov_se_fat multiscales with tproj support input dimension D <= 32");
      throw HipFail{ST_BAD_ARG};
    }
    return;
  }
  if (!a.big) hipLaunchKernelGGL((grad_fused_kernel<DT, 0, TS>), grid, dim3(256), 0, s, a);
  else if (a.D <= 8) hipLaunchKernelGGL((grad_fused_kernel<DT, 8, TS>), grid, dim3(256), 0, s, a);
  else if (a.D <= 32) hipLaunchKernelGGL((grad_fused_kernel<DT, 32, TS>), grid, dim3(256), 0, s, a);
  else if (a.D <= 64) hipLaunchKernelGGL((grad_fused_kernel<DT, 64, TS>), grid, dim3(256), 0, s, a);
  else {
    set_error("gprhip: Cov_se_fat input dimension D > 64 is not supported by the gradient kernel");
    throw HipFail{ST_BAD_ARG};
  }
}

template <typename TS>
void launch_grad_fused(const GradArgs<TS>& a, hipStream_t s) {
  dim3 grid((a.mp + 255) / 256, (a.rows + a.slab - 1) / a.slab);
  if (a.d <= 4) grad_dispatch_big<4, TS>(a, grid, s);
  else if (a.d <= 8) grad_dispatch_big<8, TS>(a, grid, s);
  else if (a.d <= 16) grad_dispatch_big<16, TS>(a, grid, s);
  else if (a.d <= 32) grad_dispatch_big<32, TS>(a, grid, s);
  else if (a.d <= 64) grad_dispatch_big<64, TS>(a, grid, s);
  else {
    set_error("gprhip: input dimension d > 64 is not supported by the gradient kernel");
    throw HipFail{ST_BAD_ARG};
  }
  GPR_HIP(hipGetLastError());
}
template void launch_grad_fused<double>(const GradArgs<double>&, hipStream_t);
template void launch_grad_fused<float>(const GradArgs<float>&, hipStream_t);

}  // namespace gprhip
