// Covariance builders for Cov_se_iso / Cov_se_fat (projection-only) -- HBM/transcendental-bound.
//
// Squared distances are accumulated exactly like the reference's scalar loops
// (lib/cov_se_iso.ml:128-144, lib/cov_se_fat.ml:232-240): sum over dimensions in
// increasing order of (x_i - z_i)*(x_i - z_i), separate multiply and add (no FMA
// contraction), then exp(log_sf2 + inv_ell2_05 * r2).  Only the last-ulp behaviour of
// exp() can differ from the CPU.
#include <cstdlib>
#include "kernels.h"
#include "exp_fast.h"

namespace gprhip {

#pragma clang fp contract(off)

// Thread <-> inducing column (coalesced stores along a row of K_nm); the block's 32 training points
// are staged in LDS zero-padded to DT dimensions, so the distance loop is branch-free and reads its
// x values as LDS broadcasts.  Padded dimensions add (0-0)^2 = 0 exactly: the rounding sequence of the
// real dimensions is unchanged.
template <int DT, typename TS>
__global__ __launch_bounds__(256) void cov_cross_kernel(CovParams cp, const double* __restrict__ pts,
                                                        int rows, int rows_p,
                                                        const double* __restrict__ Z, int m, int mp,
                                                        int d, TS* __restrict__ K) {
  const ExpK ek = exp_consts();
  __shared__ double xs[32][DT];
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int r0 = blockIdx.y * 32;
  for (int idx = threadIdx.x; idx < 32 * DT; idx += 256) {
    const int r = idx / DT, k = idx % DT;
    xs[r][k] = (k < d && r0 + r < rows) ? pts[(int64_t)(r0 + r) * d + k] : 0.0;
  }
  __syncthreads();
  if (j >= mp) return;
  double z[DT];
#pragma unroll
  for (int k = 0; k < DT; ++k) z[k] = (k < d && j < m) ? Z[(int64_t)j * d + k] : 0.0;
  const bool live_col = j < m;
  const int nr = min(32, rows_p - r0);
  for (int i = 0; i < nr; ++i) {
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < DT; ++k) {
      const double diff = xs[i][k] - z[k];
      // fp64 results follow the reference's rounding sequence (separate multiply and add); the fp32-bulk mode rounds
      // K to fp32 on store, so there the sum may use one fused multiply-add per dimension (2/3 of the instructions)
      if constexpr (sizeof(TS) == 4) acc = __builtin_fma(diff, diff, acc);
      else acc = acc + diff * diff;
    }
    const double val = (r0 + i < rows && live_col) ? exp_fast(cp.log_sf2 + cp.inv_ell2_05 * acc, ek) : 0.0;
    K[(int64_t)(r0 + i) * mp + j] = (TS)val;
  }
}

// Multiscale variants (lib/cov_se_fat.ml:102-103, :115-134, :241-251): per inducing point and dimension a
// scale ms >= 0.5; the exponent accumulates diff*(diff/scale) + log(scale).
template <int DT, typename TS>
__global__ __launch_bounds__(256) void cov_cross_ms_kernel(CovParams cp, const double* __restrict__ pts,
                                                           int rows, int rows_p,
                                                           const double* __restrict__ Z, int m, int mp,
                                                           int d, TS* __restrict__ K) {
  const ExpK ek = exp_consts();
  __shared__ double xs[32][DT];
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int r0 = blockIdx.y * 32;
  for (int idx = threadIdx.x; idx < 32 * DT; idx += 256) {
    const int r = idx / DT, k = idx % DT;
    xs[r][k] = (k < d && r0 + r < rows) ? pts[(int64_t)(r0 + r) * d + k] : 0.0;
  }
  __syncthreads();
  if (j >= mp) return;
  double z[DT], sc[DT], lsc[DT];
#pragma unroll
  for (int k = 0; k < DT; ++k) {
    const bool live = (k < d && j < m);
    z[k] = live ? Z[(int64_t)j * d + k] : 0.0;
    sc[k] = live ? cp.ms[(int64_t)j * d + k] : 1.0;
    lsc[k] = log(sc[k]);
  }
  const bool live_col = j < m;
  const int nr = min(32, rows_p - r0);
  for (int i = 0; i < nr; ++i) {
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < DT; ++k) {
      const double diff = xs[i][k] - z[k];
      acc = (acc + diff * (diff / sc[k])) + lsc[k];
    }
    const double val = (r0 + i < rows && live_col) ? exp_fast(cp.log_sf2 + cp.inv_ell2_05 * acc, ek) : 0.0;
    K[(int64_t)(r0 + i) * mp + j] = (TS)val;
  }
}

template <int DT>
__global__ __launch_bounds__(256) void cov_upper_ms_kernel(CovParams cp, const double* __restrict__ Z,
                                                           int m, int mp, int d, double jitter,
                                                           const double* __restrict__ het,
                                                           double* __restrict__ km,
                                                           double* __restrict__ kj, int rb) {
  const ExpK ek = exp_consts();
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= mp) return;
  const int r0 = blockIdx.y * rb;
  double z[DT], msc[DT];
#pragma unroll
  for (int k = 0; k < DT; ++k) {
    z[k] = (k < d && c < m) ? Z[(int64_t)c * d + k] : 0.0;
    msc[k] = (k < d && c < m) ? cp.ms[(int64_t)c * d + k] : 1.0;
  }
  for (int i = 0; i < rb; ++i) {
    const int r = r0 + i;
    if (r >= mp) break;
    double val = 0.0, valj = 0.0;
    if (r < m && c < m) {
      double acc = 0.0;
      if (r == c) {
#pragma unroll
        for (int k = 0; k < DT; ++k)
          if (k < d) acc = acc + log(msc[k] + msc[k] - 1.0);   // lib/cov_se_fat.ml:128-131
        val = exp_fast(cp.log_sf2 + cp.inv_ell2_05 * acc, ek);
        valj = (het ? val + het[c] : val) + jitter;
      } else {
        const double* x = Z + (int64_t)r * d;
        const double* msr = cp.ms + (int64_t)r * d;
#pragma unroll
        for (int k = 0; k < DT; ++k) {
          if (k < d) {
            const double diff = x[k] - z[k];
            const double scale = (msr[k] + msc[k]) - 1.0;
            acc = (acc + diff * (diff / scale)) + log(scale);
          }
        }
        val = exp_fast(cp.log_sf2 + cp.inv_ell2_05 * acc, ek);
        valj = val;
      }
    } else if (r == c) {
      valj = 1.0;
    }
    km[(int64_t)r * mp + c] = val;
    kj[(int64_t)r * mp + c] = valj;
  }
}

template <int DT>
__global__ __launch_bounds__(256) void cov_upper_kernel(CovParams cp, const double* __restrict__ Z,
                                                        int m, int mp, int d, double jitter,
                                                        const double* __restrict__ het,
                                                        double* __restrict__ km,
                                                        double* __restrict__ kj, int rb) {
  const ExpK ek = exp_consts();
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= mp) return;
  const int r0 = blockIdx.y * rb;
  double z[DT];
#pragma unroll
  for (int k = 0; k < DT; ++k) z[k] = (k < d && c < m) ? Z[(int64_t)c * d + k] : 0.0;
  for (int i = 0; i < rb; ++i) {
    const int r = r0 + i;
    if (r >= mp) break;
    double val = 0.0, valj = 0.0;
    if (r < m && c < m) {
      if (r == c) {
        val = cp.sf2;  // lib/cov_se_iso.ml:82, lib/cov_se_fat.ml:98
        valj = (het ? cp.sf2 + het[c] : cp.sf2) + jitter;  // hetero first, then jitter (fitc_gp.ml:54-55)
      } else {
        const double* x = Z + (int64_t)r * d;
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < DT; ++k) {
          if (k < d) {
            double diff = z[k] - x[k];
            acc = acc + diff * diff;
          }
        }
        val = exp_fast(cp.log_sf2 + cp.inv_ell2_05 * acc, ek);
        valj = val;
      }
    } else if (r == c) {
      valj = 1.0;  // padding: identity block, contributes log 1 = 0 to every determinant
    }
    km[(int64_t)r * mp + c] = val;
    kj[(int64_t)r * mp + c] = valj;
  }
}

// ---- point dimensions above 64 ("wide"): the dimension loop runs in chunks of 64 staged through LDS, the inducing
// coordinates come from memory (cache-resident: 16 consecutive k share a line) instead of a register array.  Same
// accumulation order as above (dimensions increasing, separate multiply and add).  No multiscales on this path.
template <typename TS>
__global__ __launch_bounds__(256) void cov_cross_wide_kernel(CovParams cp, const double* __restrict__ pts,
                                                             int rows, int rows_p,
                                                             const double* __restrict__ Z, int m, int mp,
                                                             int d, TS* __restrict__ K) {
  const ExpK ek = exp_consts();
  __shared__ double xs[32][64];
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int r0 = blockIdx.y * 32;
  const bool live_col = j < m;
  const double* zc = Z + (int64_t)min(j, m - 1) * d;
  double acc[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = 0.0;
  for (int k0 = 0; k0 < d; k0 += 64) {
    const int kc = min(64, d - k0);
    __syncthreads();
    for (int idx = threadIdx.x; idx < 32 * 64; idx += 256) {
      const int r = idx / 64, k = idx % 64;
      xs[r][k] = (k < kc && r0 + r < rows) ? pts[(int64_t)(r0 + r) * d + k0 + k] : 0.0;
    }
    __syncthreads();
    for (int k = 0; k < kc; ++k) {
      const double z = zc[k0 + k];
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        const double diff = xs[i][k] - z;
        acc[i] = acc[i] + diff * diff;
      }
    }
  }
  if (j >= mp) return;
  const int nr = min(32, rows_p - r0);
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    if (i < nr) {
      const double val = (r0 + i < rows && live_col) ? exp_fast(cp.log_sf2 + cp.inv_ell2_05 * acc[i], ek) : 0.0;
      K[(int64_t)(r0 + i) * mp + j] = (TS)val;
    }
  }
}

__global__ __launch_bounds__(256) void cov_upper_wide_kernel(CovParams cp, const double* __restrict__ Z, int m,
                                                             int mp, int d, double jitter,
                                                             const double* __restrict__ het,
                                                             double* __restrict__ km, double* __restrict__ kj) {
  const ExpK ek = exp_consts();
  __shared__ double xs[32][64];
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int r0 = blockIdx.y * 32;
  const double* zc = Z + (int64_t)min(c, m - 1) * d;
  double acc[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = 0.0;
  for (int k0 = 0; k0 < d; k0 += 64) {
    const int kc = min(64, d - k0);
    __syncthreads();
    for (int idx = threadIdx.x; idx < 32 * 64; idx += 256) {
      const int r = idx / 64, k = idx % 64;
      xs[r][k] = (k < kc && r0 + r < m) ? Z[(int64_t)(r0 + r) * d + k0 + k] : 0.0;
    }
    __syncthreads();
    for (int k = 0; k < kc; ++k) {
      const double z = zc[k0 + k];
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        const double diff = z - xs[i][k];
        acc[i] = acc[i] + diff * diff;
      }
    }
  }
  if (c >= mp) return;
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const int r = r0 + i;
    if (r < mp) {
      double val = 0.0, valj = 0.0;
      if (r < m && c < m) {
        if (r == c) {
          val = cp.sf2;
          valj = (het ? cp.sf2 + het[c] : cp.sf2) + jitter;
        } else {
          val = exp_fast(cp.log_sf2 + cp.inv_ell2_05 * acc[i], ek);
          valj = val;
        }
      } else if (r == c) {
        valj = 1.0;
      }
      km[(int64_t)r * mp + c] = val;
      kj[(int64_t)r * mp + c] = valj;
    }
  }
}


// P[r][small] = sum_big tproj(big,small) * X[r][big]  -- dgemm ~transa:`T tproj inputs (lib/cov_se_fat.ml:215-218), on
// the matrix cores for any D and d: one wavefront per 16 x 16 output tile, v_mfma_f64_16x16x4_f64 over `big` in steps
// of 4, operands straight from memory (the whole problem is 2 n D d flops: ~3e10 at n = 1M, D = 200, d = 80).
__global__ __launch_bounds__(256) void project_mfma_kernel(const double* __restrict__ X, int64_t n, int D, int d,
                                                           const double* __restrict__ tproj,
                                                           double* __restrict__ P) {
  typedef double d4 __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * 16;
  if (r0 >= n) return;
  const int64_t row = min(r0 + l15, n - 1);
  for (int c0 = 0; c0 < d; c0 += 16) {
    const int col = min(c0 + l15, d - 1);
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < D; k0 += 4) {
      const int k = k0 + lq;
      const double a = (k < D) ? X[row * D + k] : 0.0;                       // A[i = l15][k = lq]
      const double b = (k < D) ? tproj[(int64_t)col * D + k] : 0.0;          // B[k = lq][j = l15]
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {                                            // C[(l >> 4) + 4 r][l & 15]
      const int64_t orow = r0 + lq + 4 * r;
      if (orow < n && c0 + l15 < d) P[orow * d + c0 + l15] = acc[r];
    }
  }
}


template <typename F>
static void dispatch_dt(int d, F&& f) {
  if (d <= 4) f(std::integral_constant<int, 4>{});
  else if (d <= 8) f(std::integral_constant<int, 8>{});
  else if (d <= 16) f(std::integral_constant<int, 16>{});
  else if (d <= 32) f(std::integral_constant<int, 32>{});
  else if (d <= 64) f(std::integral_constant<int, 64>{});
  else f(std::integral_constant<int, 0>{});  // wide: dimension loop in chunks (no multiscales)
}
static void no_wide_multiscales(const CovParams& cp, int d) {
  if (cp.ms && d > 64) {
    set_error("gprhip: Cov_se_fat multiscales support kernel-space dimension d <= 64");
    throw HipFail{ST_BAD_ARG};
  }
}

void launch_cov_upper(const CovParams& cp, const double* Z, int m, int mp, int d, double jitter,
                      const double* het, double* km, double* kj, hipStream_t s) {
  dim3 grid(mp / 256 + (mp % 256 ? 1 : 0), (mp + 31) / 32);
  // the m x m covariance sits on the latency chain in front of the factorisation: 8 rows per thread instead of 32 puts
  // four times as many workgroups on the chip (m = 2048: 48 -> 17 us); round 5: 2 rows per thread up to m = 1024
  // (m = 512: 17.9 us as 128 workgroups of 8-row threads, profiles/r05_timeline_n50000_m512.txt)
  static const int rb_env = [] { const char* e = getenv("GPRHIP_COV_UPPER_ROWS"); return e ? atoi(e) : 0; }();
  const int rb = rb_env > 0 ? rb_env : (mp <= 1024 ? 2 : 8);
  dim3 grid8(grid.x, (mp + rb - 1) / rb);
  no_wide_multiscales(cp, d);
  dispatch_dt(d, [&](auto dt) {
    constexpr int DT = decltype(dt)::value;
    if constexpr (DT == 0)
      hipLaunchKernelGGL(cov_upper_wide_kernel, grid, dim3(256), 0, s, cp, Z, m, mp, d, jitter, het, km, kj);
    else if (cp.ms)
      hipLaunchKernelGGL((cov_upper_ms_kernel<DT>), grid8, dim3(256), 0, s, cp, Z, m, mp, d, jitter, het, km, kj, rb);
    else
      hipLaunchKernelGGL((cov_upper_kernel<DT>), grid8, dim3(256), 0, s, cp, Z, m, mp, d, jitter, het, km, kj, rb);
  });
  GPR_HIP(hipGetLastError());
}

// fp32-bulk mode, 16..64 point dimensions: the distance through the matrix cores.  K is rounded to fp32 on store in
// that mode, so the rounding sequence of the reference's direct differences is not preserved anyway, and above 16
// dimensions the scalar kernel is bound by its 3 d fp64 instructions per element:
//   |p - z|^2 = |p|^2 + |z|^2 - 2 S,  S = P Z^T as v_mfma_f64_16x16x4_f64 tiles, both sides shifted by the centroid of
//   the inducing points first (the expansion then loses digits only relative to the spread of the data).
// Workgroup: 4 wavefronts x 32 columns, a slab of 256 rows staged through LDS 64 rows at a time (as grad_mfma.hip).
typedef double cd4 __attribute__((ext_vector_type(4)));
template <int KS4, typename TS>
__global__ __launch_bounds__(256) void cov_cross_mfma_kernel(CovParams cp, const double* __restrict__ pts, int rows,
                                                             int rows_p, const double* __restrict__ Z, int m, int mp,
                                                             int d, const double* __restrict__ shift,
                                                             TS* __restrict__ K) {
  constexpr int DP = KS4 * 4, LDP = DP + 1, RC = 64, SLAB = 256;
  const ExpK ek = exp_consts();
  __shared__ double ps[RC * LDP];
  __shared__ double pn[RC];
  __shared__ double sh[DP];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid < DP) sh[tid] = tid < d ? shift[tid] : 0.0;
  __syncthreads();
  const int l15 = lane & 15, lq = lane >> 4;
  const int cb = blockIdx.x * 128 + wv * 32;
  double zf[2][KS4], zn[2];
  bool live_c[2];
#pragma unroll
  for (int jt = 0; jt < 2; ++jt) {
    const int col = cb + jt * 16 + l15;
    live_c[jt] = col < m;
    double s2 = 0.0;
#pragma unroll
    for (int s = 0; s < KS4; ++s) {
      const int k = 4 * s + lq;
      const double z = (live_c[jt] && k < d) ? Z[(int64_t)col * d + k] - sh[k] : 0.0;
      zf[jt][s] = z;
      s2 = __builtin_fma(z, z, s2);
    }
    s2 += __shfl_xor(s2, 16);
    s2 += __shfl_xor(s2, 32);
    zn[jt] = s2;
  }
  const int r0 = blockIdx.y * SLAB, r1 = min(rows_p, r0 + SLAB);
  for (int rb = r0; rb < r1; rb += RC) {
    __syncthreads();
    for (int idx = tid; idx < RC * DP; idx += 256) {
      const int r = idx / DP, k = idx % DP;
      ps[r * LDP + k] = (k < d && rb + r < rows) ? pts[(int64_t)(rb + r) * d + k] - sh[k] : 0.0;
    }
    __syncthreads();
    if (tid < RC) {
      double s2 = 0.0;
      for (int k = 0; k < DP; ++k) s2 = __builtin_fma(ps[tid * LDP + k], ps[tid * LDP + k], s2);
      pn[tid] = s2;
    }
    __syncthreads();
    for (int rt = 0; rt < RC / 16; ++rt) {
      if (rb + rt * 16 >= r1) break;
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) {
        cd4 s4 = (cd4){0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < KS4; ++s)
          s4 = __builtin_amdgcn_mfma_f64_16x16x4f64(ps[(rt * 16 + l15) * LDP + 4 * s + lq], zf[jt][s], s4, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int lr = rt * 16 + lq + 4 * r, row = rb + lr;
          const double dist = fmax(__builtin_fma(-2.0, s4[r], pn[lr] + zn[jt]), 0.0);
          const double kv = exp_fast(__builtin_fma(cp.inv_ell2_05, dist, cp.log_sf2), ek);
          if (row < r1) K[(int64_t)row * mp + cb + jt * 16 + l15] = (TS)((row < rows && live_c[jt]) ? kv : 0.0);
        }
      }
    }
  }
}

template <typename TS>
void launch_cov_cross(const CovParams& cp, const double* pts, int rows, int rows_p, const double* Z,
                      int m, int mp, int d, TS* K, hipStream_t s, const double* shift) {
  no_wide_multiscales(cp, d);
  if constexpr (sizeof(TS) == 4) {
    if (shift && !cp.ms && d >= 16 && d <= 64 && mp % 128 == 0) {
      dim3 g2(mp / 128, (rows_p + 255) / 256);
      if (d <= 16) hipLaunchKernelGGL((cov_cross_mfma_kernel<4, TS>), g2, dim3(256), 0, s, cp, pts, rows, rows_p, Z, m, mp, d, shift, K);
      else if (d <= 32) hipLaunchKernelGGL((cov_cross_mfma_kernel<8, TS>), g2, dim3(256), 0, s, cp, pts, rows, rows_p, Z, m, mp, d, shift, K);
      else hipLaunchKernelGGL((cov_cross_mfma_kernel<16, TS>), g2, dim3(256), 0, s, cp, pts, rows, rows_p, Z, m, mp, d, shift, K);
      GPR_HIP(hipGetLastError());
      return;
    }
  }
  dim3 grid(mp / 256 + (mp % 256 ? 1 : 0), (rows_p + 31) / 32);
  dispatch_dt(d, [&](auto dt) {
    constexpr int DT = decltype(dt)::value;
    if constexpr (DT == 0)
      hipLaunchKernelGGL((cov_cross_wide_kernel<TS>), grid, dim3(256), 0, s, cp, pts, rows, rows_p, Z, m, mp, d, K);
    else if (cp.ms)
      hipLaunchKernelGGL((cov_cross_ms_kernel<DT, TS>), grid, dim3(256), 0, s, cp, pts, rows, rows_p, Z, m, mp, d, K);
    else
      hipLaunchKernelGGL((cov_cross_kernel<DT, TS>), grid, dim3(256), 0, s, cp, pts, rows, rows_p, Z, m, mp, d, K);
  });
  GPR_HIP(hipGetLastError());
}
template void launch_cov_cross<double>(const CovParams&, const double*, int, int, const double*, int, int, int,
                                       double*, hipStream_t, const double*);
template void launch_cov_cross<float>(const CovParams&, const double*, int, int, const double*, int, int, int,
                                      float*, hipStream_t, const double*);

void launch_project(const double* X, int64_t n, int D, int d, const double* tproj, double* P,
                    hipStream_t s) {
  hipLaunchKernelGGL(project_mfma_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, s, X, n, D, d, tproj, P);
  GPR_HIP(hipGetLastError());
}

}  // namespace gprhip
