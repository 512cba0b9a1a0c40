// Fused gradient pass on the matrix cores (Cov_se_iso, Cov_se_fat without multiscales).
//
// Same outputs as grad_fused_kernel (rowops.hip): with E = X .* K_nm (K_nm recomputed, never stored)
//   colpart[slab][0][c]       = sum_r E_rc                    `Factor / Log_sf2, lib/fitc_gp.ml:991
//   colpart[slab][1+k][c]     = sum_r p_kr E_rc               `Sparse_cols inducing derivative, lib/cov_se_iso.ml:301-327
//   colpart[slab][1+d+k][c]   = sum_r x_big,kr E_rc           `Proj, lib/cov_se_fat.ml:570-596
//   scalpart[...][0..1]       = sum E, sum E * |p_r - z_c|^2  Log_ell, lib/cov_se_iso.ml:313-318
// but the three contractions over the point dimension run as v_mfma_f64_16x16x4_f64:
//   S = P Z^T  (the "distance GEMM" of SURVEY 8(d)):  |p_r - z_c|^2 = |p_r|^2 + |z_c|^2 - 2 S_rc
//     (both sides are first shifted by the centroid of the inducing points, so the expansion loses digits only
//      relative to the spread of the data, not to a common offset; sum_r p_kr E_rc is corrected by shift_k * sum_r E_rc)
//   G += P^T E, Gb += X_big^T E  (the "inducing-gradient GEMM"); with projection hypers only Gb is accumulated --
//     P = X_big tproj, so G = tproj^T Gb follows from the reduced sums (launch_proj_inducing_grad, once per pass)
//     and a third of the kernel's MFMAs goes away
// A 16x16 tile of E comes out of the elementwise step in the accumulator layout (lane holds rows lq + 4r of column
// l15), which is exactly four B operands (k = lq) of the next MFMAs: E never leaves the registers.
// Workgroup: 4 wavefronts x 32 columns, one slab of a.slab rows, rows staged through LDS 32 at a time.
// KR ("K resident", Cov_se_fat with projection hypers): K_nm of pass 1 is still in memory (gprhip.hip keeps it when the
// device has room), so E = X .* K is read -- no distance product, no exp: the kernel is left with the X_big^T E
// MFMAs and 2 x 4 (fp32-bulk) or 2 x 8 bytes per element of HBM traffic (C3: 24 -> 9 ms per evaluation).  Cov_se_fat
// has no length-scale hyper, so the sum E .* D of the iso kernel is not needed there.
#include "kernels.h"
#include "exp_fast.h"

namespace gprhip {

namespace {

typedef double gd4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ gd4 mfma4(double a, double b, gd4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ double wsum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

constexpr int G_RC = 64;     // rows per staged chunk

// KS4 = ceil(d / 4) k-steps of the distance product, DT = ceil(d / 16) tiles of point dimensions,
// BT = ceil(D / 16) tiles of original input dimensions (0: no projection hypers)
template <int KS4, int DT, int BT, typename TS, bool KR = false>
__global__ __launch_bounds__(256, (DT + BT <= 4 && KS4 <= 8) ? 2 : 1) void grad_mfma_kernel(GradArgs<TS> a) {
  constexpr int DP = DT * 16, LDP = DP + 1;
  constexpr int BP = BT > 0 ? BT * 16 : 1, LDB = BP + 1;
  // Up to 32 point dimensions and 32 input dimensions: the staged points are double-buffered -- chunk c+1 is fetched
  // into registers while chunk c is consumed, one barrier per chunk -- and the X values of the next row tile are
  // loaded while the current one is computed.  The wider instantiations have no registers (and no LDS) to spare for
  // that and keep one staging buffer.
  constexpr bool PF = (DT <= 2 && BT <= 2);
  constexpr int NBUF = PF ? 2 : 1;
  __shared__ double ps[KR ? 1 : NBUF * G_RC * LDP];
  __shared__ double bs[BT > 0 ? NBUF * G_RC * LDB : 1];
  __shared__ double pn[KR ? 1 : NBUF * G_RC];
  __shared__ double red[4][2];
  __shared__ double sh[DP];  // the expansion offset, zero-padded
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid < DP) sh[tid] = (a.shift && tid < a.d) ? a.shift[tid] : 0.0;
  __syncthreads();
  const int l15 = lane & 15, lq = lane >> 4;
  const int cb = blockIdx.x * 128 + wv * 32;
  const ExpK ek = exp_consts();

  double zf[2][KS4], zn[2];
  bool live_c[2];
#pragma unroll
  for (int jt = 0; jt < 2; ++jt) {
    const int col = cb + jt * 16 + l15;
    live_c[jt] = col < a.m;
    double s2 = 0.0;
    if constexpr (!KR) {
#pragma unroll
      for (int s = 0; s < KS4; ++s) {
        const int k = 4 * s + lq;
        const double z = (live_c[jt] && k < a.d) ? a.Z[(int64_t)col * a.d + k] - sh[k] : 0.0;
        zf[jt][s] = z;
        s2 += z * z;
      }
      s2 += __shfl_xor(s2, 16);
      s2 += __shfl_xor(s2, 32);
    }
    zn[jt] = s2;
  }
  gd4 g[DT][2], gb[BT > 0 ? BT : 1][2];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) g[t][jt] = (gd4){0, 0, 0, 0};
#pragma unroll
  for (int t = 0; t < (BT > 0 ? BT : 1); ++t)
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) gb[t][jt] = (gd4){0, 0, 0, 0};
  double cs[2] = {0.0, 0.0}, sE = 0.0, sED = 0.0;

  const int r0 = blockIdx.y * a.slab;
  const int r1 = min(a.rows, r0 + a.slab);

  // X values of one 16-row tile (rows first .. first+15) for this wave's 32 columns
  // (KR: times the K values of the same elements -- the loaded tile is E itself)
  auto load_x = [&](int first, double (&xv)[2][4]) {
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = first + lq + 4 * r;
        const int64_t off = (int64_t)row * a.mp + cb + jt * 16 + l15;
        if constexpr (KR) xv[jt][r] = (row < r1) ? (double)a.X[off] * (double)a.K[off] : 0.0;
        else xv[jt][r] = (row < r1) ? (double)a.X[off] : 0.0;
      }
  };
  // one 16-row tile: rows rb + 16 rt ... of the chunk staged at psb / bsb / pnb
  auto tile = [&](const double* psb, const double* bsb, const double* pnb, int rb, int rt, const double (&xv)[2][4]) {
    double ev[2][4];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
      if constexpr (KR) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {  // (padded rows load zero; padded columns are masked, whatever K holds there)
          const double e = live_c[jt] ? xv[jt][r] : 0.0;
          ev[jt][r] = e;
          cs[jt] += e;
          sE += e;
        }
      } else {
        gd4 s4 = (gd4){0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < KS4; ++s) s4 = mfma4(psb[(rt * 16 + l15) * LDP + 4 * s + lq], zf[jt][s], s4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int lr = rt * 16 + lq + 4 * r;
          const double dist = fmax(pnb[lr] + zn[jt] - 2.0 * s4[r], 0.0);
          const double kv = exp_fast(a.log_sf2 + a.inv_ell2_05 * dist, ek);
          const double e = (live_c[jt] && rb + lr < r1) ? xv[jt][r] * kv : 0.0;
          ev[jt][r] = e;
          cs[jt] += e;
          sE += e;
          sED += e * dist;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int prow = (rt * 16 + 4 * r + lq);
      if (BT == 0) {
#pragma unroll
        for (int t = 0; t < DT; ++t) {
          const double ap = psb[prow * LDP + t * 16 + l15];
          g[t][0] = mfma4(ap, ev[0][r], g[t][0]);
          g[t][1] = mfma4(ap, ev[1][r], g[t][1]);
        }
      }
      if (BT > 0) {
#pragma unroll
        for (int t = 0; t < (BT > 0 ? BT : 1); ++t) {
          const double ab = bsb[prow * LDB + t * 16 + l15];
          gb[t][0] = mfma4(ab, ev[0][r], gb[t][0]);
          gb[t][1] = mfma4(ab, ev[1][r], gb[t][1]);
        }
      }
    }
  };

  if constexpr (PF) {
    // thread -> (row tid/DP + (256/DP) j, dimension tid%DP) of a chunk: the DP lanes of a row also reduce its squared
    // norm; the original inputs (projection hypers) are staged the same way with BP lanes per row
    constexpr int RPP = 256 / DP, NPV = G_RC / RPP;
    constexpr int RPB = 256 / BP, NBV = BT > 0 ? G_RC / RPB : 1;
    const int sr = tid / DP, sk = tid % DP;
    const int br_ = tid / BP, bk = tid % BP;
    double pv[NPV], bv[NBV];
    auto fetch_pts = [&](int rb) {
      if constexpr (!KR) {  // (K resident: neither distances nor the P^T E products need the points)
#pragma unroll
        for (int j = 0; j < NPV; ++j) {
          const int row = rb + sr + RPP * j;
          pv[j] = (sk < a.d && row < r1) ? a.pts[(int64_t)row * a.d + sk] - sh[sk] : 0.0;
        }
      }
      if constexpr (BT > 0) {
#pragma unroll
        for (int j = 0; j < NBV; ++j) {
          const int row = rb + br_ + RPB * j;
          bv[j] = (bk < a.D && row < r1) ? a.big[(int64_t)row * a.D + bk] : 0.0;
        }
      }
    };
    auto store_pts = [&](int buf) {
      if constexpr (!KR) {
#pragma unroll
        for (int j = 0; j < NPV; ++j) {
          const int r = sr + RPP * j;
          ps[buf * G_RC * LDP + r * LDP + sk] = pv[j];
          double s2 = pv[j] * pv[j];
          s2 += __shfl_xor(s2, 1);
          s2 += __shfl_xor(s2, 2);
          s2 += __shfl_xor(s2, 4);
          s2 += __shfl_xor(s2, 8);
          if constexpr (DP == 32) s2 += __shfl_xor(s2, 16);
          if (sk == 0) pn[buf * G_RC + r] = s2;
        }
      }
      if constexpr (BT > 0) {
#pragma unroll
        for (int j = 0; j < NBV; ++j) bs[buf * G_RC * LDB + (br_ + RPB * j) * LDB + bk] = bv[j];
      }
    };
    double xv[2][4], xn[2][4];
    fetch_pts(r0);
    load_x(r0, xv);
    store_pts(0);
    __syncthreads();
    int buf = 0;
    for (int rb = r0; rb < r1; rb += G_RC, buf ^= 1) {
      const bool more = rb + G_RC < r1;
      if (more) fetch_pts(rb + G_RC);
#pragma unroll 1
      for (int rt = 0; rt < G_RC / 16; ++rt) {
        if (rb + rt * 16 >= r1) break;
        load_x(rb + rt * 16 + 16, xn);  // rows beyond the slab load nothing
        tile(ps + buf * G_RC * LDP, bs + (BT > 0 ? buf * G_RC * LDB : 0), pn + buf * G_RC, rb, rt, xv);
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
          for (int r = 0; r < 4; ++r) xv[jt][r] = xn[jt][r];
      }
      if (more) store_pts(buf ^ 1);
      __syncthreads();
    }
  } else {
    for (int rb = r0; rb < r1; rb += G_RC) {
      __syncthreads();
      if constexpr (!KR) {
        for (int idx = tid; idx < G_RC * DP; idx += 256) {
          const int r = idx / DP, k = idx % DP;
          ps[r * LDP + k] = (k < a.d && rb + r < r1) ? a.pts[(int64_t)(rb + r) * a.d + k] - sh[k] : 0.0;
        }
      }
      if (BT > 0) {
        for (int idx = tid; idx < G_RC * BP; idx += 256) {
          const int r = idx / BP, k = idx % BP;
          bs[r * LDB + k] = (k < a.D && rb + r < r1) ? a.big[(int64_t)(rb + r) * a.D + k] : 0.0;
        }
      }
      __syncthreads();
      if constexpr (!KR) {
        if (tid < G_RC) {
          double s2 = 0.0;
          for (int k = 0; k < DP; ++k) s2 += ps[tid * LDP + k] * ps[tid * LDP + k];
          pn[tid] = s2;
        }
        __syncthreads();
      }
#pragma unroll 1  // one row tile's worth of registers: two wavefronts per SIMD stay resident
      for (int rt = 0; rt < G_RC / 16; ++rt) {
        if (rb + rt * 16 >= r1) break;
        double xv[2][4];
        load_x(rb + rt * 16, xv);
        tile(ps, bs, pn, rb, rt, xv);
      }
    }
  }

  double* cp = a.colpart + (int64_t)blockIdx.y * a.col_rows * a.mp;
#pragma unroll
  for (int jt = 0; jt < 2; ++jt) {
    const int col = cb + jt * 16 + l15;
    double c = cs[jt];
    c += __shfl_xor(c, 16);
    c += __shfl_xor(c, 32);
    if (lq == 0) cp[col] = c;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int dim = t * 16 + lq + 4 * r;
        if (dim < a.d) cp[(int64_t)(1 + dim) * a.mp + col] = (BT == 0) ? g[t][jt][r] + sh[dim] * c : 0.0;
      }
    if (BT > 0) {
#pragma unroll
      for (int t = 0; t < (BT > 0 ? BT : 1); ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int dim = t * 16 + lq + 4 * r;
          if (dim < a.D) cp[(int64_t)(a.d + 1 + dim) * a.mp + col] = gb[t][jt][r];
        }
    }
  }
  sE = wsum(sE);
  sED = wsum(sED);
  if (lane == 0) {
    red[wv][0] = sE;
    red[wv][1] = sED;
  }
  __syncthreads();
  if (tid < 2)
    a.scalpart[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 + tid] =
        (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

template <int KS4, int DT, typename TS>
void dispatch_big(const GradArgs<TS>& a, dim3 grid, hipStream_t s) {
  if (a.K && a.big) {  // K resident (projection hypers only: the caller passes K for no other launch)
    if (a.D <= 16) hipLaunchKernelGGL((grad_mfma_kernel<KS4, DT, 1, TS, true>), grid, dim3(256), 0, s, a);
    else if (a.D <= 32) hipLaunchKernelGGL((grad_mfma_kernel<KS4, DT, 2, TS, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((grad_mfma_kernel<KS4, DT, 4, TS, true>), grid, dim3(256), 0, s, a);
    return;
  }
  if (!a.big) hipLaunchKernelGGL((grad_mfma_kernel<KS4, DT, 0, TS>), grid, dim3(256), 0, s, a);
  else if (a.D <= 16) hipLaunchKernelGGL((grad_mfma_kernel<KS4, DT, 1, TS>), grid, dim3(256), 0, s, a);
  else if (a.D <= 32) hipLaunchKernelGGL((grad_mfma_kernel<KS4, DT, 2, TS>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((grad_mfma_kernel<KS4, DT, 4, TS>), grid, dim3(256), 0, s, a);
}

}  // namespace

// Column blocks (128 columns each) of the MFMA gradient kernel; 0 when the launch is not eligible
// (multiscales, or more than 64 point / input dimensions) and the scalar kernel has to be used.
template <typename TS>
int grad_mfma_col_blocks(const GradArgs<TS>& a) {
  if (a.ms || a.d > 64 || a.D > 64 || (a.mp % 128) != 0) return 0;
  return a.mp / 128;
}

template <typename TS>
void launch_grad_mfma(const GradArgs<TS>& a, hipStream_t s) {
  dim3 grid(a.mp / 128, (a.rows + a.slab - 1) / a.slab);
  if (a.d <= 4) dispatch_big<1, 1, TS>(a, grid, s);
  else if (a.d <= 8) dispatch_big<2, 1, TS>(a, grid, s);
  else if (a.d <= 16) dispatch_big<4, 1, TS>(a, grid, s);
  else if (a.d <= 32) dispatch_big<8, 2, TS>(a, grid, s);
  else dispatch_big<16, 4, TS>(a, grid, s);
  GPR_HIP(hipGetLastError());
}

// rows 1..d of the reduced column accumulators from rows d+1..d+D:  sum_r p_kr E_rc = sum_b tproj(b, k) sum_r x_br E_rc
__global__ __launch_bounds__(256) void proj_inducing_grad_kernel(double* __restrict__ acc, int mp, int d, int D,
                                                                 const double* __restrict__ tproj) {
  const int c = blockIdx.x * 256 + threadIdx.x, k = blockIdx.y;
  if (c >= mp) return;
  double s = 0.0;
  for (int b = 0; b < D; ++b) s += tproj[(int64_t)k * D + b] * acc[(int64_t)(d + 1 + b) * mp + c];
  acc[(int64_t)(1 + k) * mp + c] += s;
}
void launch_proj_inducing_grad(double* acc, int mp, int d, int D, const double* tproj, hipStream_t s) {
  hipLaunchKernelGGL(proj_inducing_grad_kernel, dim3((mp + 255) / 256, d), dim3(256), 0, s, acc, mp, d, D, tproj);
  GPR_HIP(hipGetLastError());
}

template int grad_mfma_col_blocks<double>(const GradArgs<double>&);
template int grad_mfma_col_blocks<float>(const GradArgs<float>&);
template void launch_grad_mfma<double>(const GradArgs<double>&, hipStream_t);
template void launch_grad_mfma<float>(const GradArgs<float>&, hipStream_t);

}  // namespace gprhip
