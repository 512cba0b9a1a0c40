// Row passes and finish stage of problems with few inducing points (m <= 64, d <= 16 point dimensions -- D <= 64 input
// dimensions in front of a projection --, any number of rows).  The reference's own shapes (n = 1000..2000, m = 10..50,
// test/save_data.ml, test/gen_data.ml) spend their time in launches, not in arithmetic -- through the engine a gradient
// evaluation is 9 contraction launches of 16-22 us each plus ~20 small kernels, 0.40 ms; and with many rows the engine
// pads m to its 128-wide tile (6.5x the flops of m = 50).  Here each pass is ONE kernel per 64-row block that keeps the block's rows of K, V, Q' and X in LDS and the 64 x 64 corners of U^-1 / R~^-1 beside
// them, plus one fixed-order reduction of the per-workgroup partial sums into the exchange buffers -- same buffers, same
// layout as the engine path writes (do_pass1 / do_pass2), so everything around the two passes is shared.
//   pass 1: K (lib/cov_se_iso.ml:128-159, lib/cov_se_fat.ml:224-240), V = K U^-1 (lib/fitc_gp.ml:226-227), r, s, 1/s
//           (:155-166, :222-223), B~ part = V^T diag(is) V, c~ part = V^T (is y)
//   pass 2: Q' = V R~^-1, q_diag, w, v (:1048, :1092-1108, :1158-1181), X~ = diag(is) Q' R~^-T - diag(v) V - w t~^T,
//           X = X~ U^-T (:931-939, :1204-1206), E = X .* K column sums (:975-1003), G~ part = V^T diag(v) V (:1198-1203)
// All products run as v_mfma_f64_16x16x4_f64 tiles on LDS operands: wavefront w owns rows 16w..16w+15 of the block.
#include <algorithm>
#include <type_traits>

#include "kernels.h"
#include "exp_fast.h"

namespace gprhip {

namespace {

constexpr int SM = 64;    // inducing points (padded) the small path handles
constexpr int SLD = 66;   // leading dimension of the 64 x 64 LDS matrices
constexpr int SRB = 64;   // training points per block iteration
constexpr int P1LEN = SM * SM + SM + 4;  // pass-1 partial of a workgroup: B~ part | c~ part | sum log s, sum y^2/s, sum r/s, -
// pass-2 partial: G~ part | column sums: E, p_k E (d), x_big E (D), with multiscales p_k^2 E (d) | `Proj term (D d) | 8 scalars
__host__ __device__ constexpr int p2len(int d, int D, int ms = 0) {
  return SM * SM + (1 + d + D + (ms ? d : 0)) * SM + D * d + 8;
}

typedef double sd4 __attribute__((ext_vector_type(4)));
// lane supplies A[lane&15][lane>>4] and B[lane>>4][lane&15]; accumulator element r is D[(lane>>4) + 4r][lane&15]
__device__ __forceinline__ sd4 mfma_f64(double a, double b, sd4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ double sum16(double v) {  // over the 16 lanes that share lane >> 4
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 8);
  return v;
}
__device__ __forceinline__ double sum64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// rows [16 wv, 16 wv + 16) of  A (LDS, [64][SLD]) times  B (LDS, [64][SLD]; TRANS: times B^T)  -> acc[ct], ct = column tile.
// Fully unrolled, all fragments of a half of the k-range loaded before its 32 MFMAs: with the loop left rolled every
// step waits for its own LDS reads and a product takes 2.5-4.5 us instead of ~2.
// (NB = k-steps per batch: 8, or 4 where registers are short)
template <bool TRANS, int NB = 8>
__device__ __forceinline__ void rows_times(const double* A, const double* B, int wv, int l15, int lq, sd4 (&acc)[4]) {
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) acc[ct] = sd4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int h = 0; h < 16 / NB; ++h) {
    double af[NB], bf[NB][4];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int kk = NB * h + j;
      af[j] = A[(16 * wv + l15) * SLD + 4 * kk + lq];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
        bf[j][ct] = TRANS ? B[(16 * ct + l15) * SLD + 4 * kk + lq] : B[(4 * kk + lq) * SLD + 16 * ct + l15];
    }
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) acc[ct] = mfma_f64(af[j], bf[j][ct], acc[ct]);
  }
}

// acc[ct] += (T^T diag(wt) T) tile (wv, ct) over the 64 rows of T (LDS)
__device__ __forceinline__ void gram_update(const double* T, const double* wt, int wv, int l15, int lq, sd4 (&acc)[4]) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    double af[8], bf[8][4];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 4 * (8 * h + j) + lq;
      af[j] = T[k * SLD + 16 * wv + l15] * wt[k];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) bf[j][ct] = T[k * SLD + 16 * ct + l15];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) acc[ct] = mfma_f64(af[j], bf[j][ct], acc[ct]);
  }
}

// sum_g part[g * stride] over the workgroups' partials, in order; sixty-four loads in flight at a time from 64 partials on
// (a batch is one memory round trip: with sixteen per batch 256 partials were sixteen dependent round trips, 26 us of a
// reduction whose ten workgroups do nothing else -- timeline at n = 10 000, m = 256), sixteen below that.  Same order of
// additions either way.
__device__ __forceinline__ double sum_parts(const double* __restrict__ part, int64_t stride, int ng) {
  double acc = 0.0;
  int g0 = 0;
  for (; g0 + 64 <= ng; g0 += 64) {
    double v[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) v[j] = part[(int64_t)(g0 + j) * stride];
#pragma unroll
    for (int j = 0; j < 64; ++j) acc += v[j];
  }
  for (; g0 < ng; g0 += 16) {
    double v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = (g0 + j < ng) ? part[(int64_t)(g0 + j) * stride] : 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc += v[j];
  }
  return acc;
}

// 64 x 64 corner of a row-major mp x mp matrix -> LDS; eight 16-byte loads per thread, all issued before the first store
__device__ __forceinline__ void load_corner(const double* __restrict__ M, int mp, double* L, int tid) {
  double2 v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int idx = tid + 256 * j, r = idx >> 5, c2 = (idx & 31) * 2;
    v[j] = *reinterpret_cast<const double2*>(M + (int64_t)r * mp + c2);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int idx = tid + 256 * j, r = idx >> 5, c2 = (idx & 31) * 2;
    *reinterpret_cast<double2*>(L + r * SLD + c2) = v[j];
  }
}

}  // namespace

// workgroups of a pass at most (each walks blocks b, b + groups, ...): what the chip holds at once -- two per CU for
// pass 1 (77 KB of LDS each), one per CU for pass 2 (150 KB)
constexpr int SMALL_GROUPS1 = 512, SMALL_GROUPS2 = 256;
int64_t small_part_len(int d, int D) {
  return std::max((int64_t)SMALL_GROUPS1 * P1LEN, (int64_t)SMALL_GROUPS2 * p2len(d, D, 1));
}
static int small_groups(int rows_p, int cap) { return std::min(cap, rows_p / SRB); }

// MS: Cov_se_fat multiscales (lib/cov_se_fat.ml:241-251; a.cp.ms = exp(log_multiscales_m05) + 1/2 as [mp][d]): the exponent
// accumulates diff * (diff / scale) + log(scale) per dimension, as cov_cross_ms_kernel
template <int DT, bool MS>
__global__ __launch_bounds__(256) void small_pass1_kernel(SmallPass1Args a) {
  extern __shared__ __attribute__((aligned(16))) double small_lds[];
  double* const Ui = small_lds;         // [SM][SLD]  U^-1
  double* const Kt = Ui + SM * SLD;     // [SRB][SLD] K of the block, then V in place
  double* const xs = Kt + SRB * SLD;    // [SRB][DT]
  double* const isr = xs + SRB * DT;    // [SRB] 1/s
  double* const yisr = isr + SRB;       // [SRB] y/s
  double* const rs = yisr + SRB;        // [SRB] rowsum(V.^2)
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, lq = lane >> 4;
  const ExpK ek = exp_consts();
  load_corner(a.uinv, a.mp, Ui, tid);
  const int col = lane, rg = wv;  // covariance / column-sum phases: thread = (column, group of 16 rows)
  const bool live_c = col < a.m;
  double z[DT], sc[MS ? DT : 1], lsc[MS ? DT : 1];
#pragma unroll
  for (int k = 0; k < DT; ++k) {
    z[k] = (k < a.d && live_c) ? a.Z[(int64_t)col * a.d + k] : 0.0;
    if constexpr (MS) {
      sc[k] = (k < a.d && live_c) ? a.cp.ms[(int64_t)col * a.d + k] : 1.0;
      lsc[k] = log(sc[k]);
    }
  }
  sd4 accB[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) accB[ct] = sd4{0.0, 0.0, 0.0, 0.0};
  double csum = 0.0, p_log = 0.0, p_y2 = 0.0, p_isr = 0.0;
  const int nblk = a.rows_p / SRB;
  for (int b = blockIdx.x; b < nblk; b += gridDim.x) {
    const int r0 = b * SRB;
    __syncthreads();
    for (int idx = tid; idx < SRB * DT; idx += 256) {
      const int r = idx / DT, k = idx % DT;
      xs[idx] = (k < a.d && r0 + r < a.rows) ? a.pts[(int64_t)(r0 + r) * a.d + k] : 0.0;
    }
    const double yreg = (tid < SRB && a.y && r0 + tid < a.rows) ? a.y[r0 + tid] : 0.0;  // used by the row phase below
    __syncthreads();
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
      const int r = rg * 16 + i;
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < DT; ++k) {  // (dimensions beyond d are zero on both sides, scale 1: they add exactly 0)
        const double diff = xs[r * DT + k] - z[k];
        if constexpr (MS) acc = (acc + diff * (diff / sc[k])) + lsc[k];
        else acc = acc + diff * diff;
      }
      const double kv = (r0 + r < a.rows && live_c) ? exp_fast(a.cp.log_sf2 + a.cp.inv_ell2_05 * acc, ek) : 0.0;
      Kt[r * SLD + col] = kv;
      if (a.Kout) a.Kout[(int64_t)(r0 + r) * SM + col] = kv;  // kept for pass 2 (E = X .* K without a second exp)
    }
    __syncthreads();
    sd4 acc[4];
    rows_times<false>(Kt, Ui, wv, l15, lq, acc);  // V = K U^-1
    double s2[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double s = 0.0;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) s += acc[ct][r] * acc[ct][r];
      s2[r] = sum16(s);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * wv + lq + 4 * r;
      if (l15 == 0) rs[row] = s2[r];
      double* vrow = a.V + (int64_t)(r0 + row) * a.mp;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        Kt[row * SLD + 16 * ct + l15] = acc[ct][r];  // rows of this wavefront only: in place
        vrow[16 * ct + l15] = acc[ct][r];
        vrow[SM + 16 * ct + l15] = 0.0;  // columns 64..127 of the padded store
      }
    }
    __syncthreads();
    if (tid < SRB) {  // r, s = r + sigma2, 1/s, sum log s  (as pass1_rows_kernel)
      const int row = r0 + tid;
      double rr = 0.0, is = 0.0, yis = 0.0;
      if (row < a.rows) {
        rr = a.cp.sf2 - rs[tid];
        const double s = rr + a.sigma2;
        is = 1.0 / s;
        const double y = yreg;
        yis = is * y;
        p_log += log(s);
        p_y2 += is * y * y;
        p_isr += is * rr;
      }
      a.r[row] = rr;
      a.is[row] = is;
      a.yis[row] = yis;
      isr[tid] = is;
      yisr[tid] = yis;
    }
    __syncthreads();
    gram_update(Kt, isr, wv, l15, lq, accB);
    for (int i = 0; i < 16; ++i) {
      const int k = rg * 16 + i;
      csum += Kt[k * SLD + col] * yisr[k];
    }
  }
  double* part = a.part + (int64_t)blockIdx.x * P1LEN;
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) part[(16 * wv + lq + 4 * r) * SM + 16 * ct + l15] = accB[ct][r];
  __syncthreads();
  Kt[rg * SLD + col] = csum;
  __syncthreads();
  if (tid < SM) part[SM * SM + tid] = (Kt[tid] + Kt[SLD + tid]) + (Kt[2 * SLD + tid] + Kt[3 * SLD + tid]);
  if (wv == 0) {
    p_log = sum64(p_log);
    p_y2 = sum64(p_y2);
    p_isr = sum64(p_isr);
    if (lane == 0) {
      part[SM * SM + SM + 0] = p_log;
      part[SM * SM + SM + 1] = p_y2;
      part[SM * SM + SM + 2] = p_isr;
      part[SM * SM + SM + 3] = 0.0;
    }
  }
}

// exchange-1 buffer from the pass-1 partials, workgroups summed in order: the (0,0) upper tile (128 x 128; zero outside
// the 64 x 64 corner), c~ (mp entries) and the scalar tail
__global__ __launch_bounds__(256) void small_reduce1_kernel(const double* __restrict__ part, int ng, int mp,
                                                            double* __restrict__ tile, double* __restrict__ cvec,
                                                            double* __restrict__ tail) {
  const int tid = threadIdx.x;
  if (blockIdx.x < TILE * TILE / 256) {
    const int idx = blockIdx.x * 256 + tid, r = idx / TILE, c = idx % TILE;
    tile[idx] = (r < SM && c < SM) ? sum_parts(part + r * SM + c, P1LEN, ng) : 0.0;
    return;
  }
  if (tid < mp) cvec[tid] = (tid < SM) ? sum_parts(part + SM * SM + tid, P1LEN, ng) : 0.0;
  else if (tid >= 192 && tid < 196) tail[tid - 192] = sum_parts(part + SM * SM + SM + (tid - 192), P1LEN, ng);
}

// DT: padded point dimension (d <= DT); DBT: padded dimension of the original inputs of a projected kernel (D <= DBT)
// MS (multiscales, d <= 8): K as grad_fused_ms_kernel forms it, the extra column sums of p_k^2 E (`Log_multiscale_m05,
// lib/cov_se_fat.ml:598-622), and for the `Proj derivative one weight per (row, dimension), sum_c E_rc / ms_kc (:585-595),
// formed from the E tile in LDS.  MS instantiations take the staged-inputs (WIDE) route for any D.
// KR: K_nm of pass 1 is read back (a.Kin, [rows_p][64]: requested at the top of a block, parked in V's tile once V is done
// with) instead of recomputed -- the sixteen exp per thread and block are a fifth of this kernel otherwise; instantiated
// for d <= 8, D <= 16, no multiscales.
template <int DT, int DBT, bool MS, bool KR = false>
__global__ __launch_bounds__(256) void small_pass2_kernel(SmallPass2Args a) {
  extern __shared__ __attribute__((aligned(16))) double small_lds[];
  double* const Ui = small_lds;          // [SM][SLD]  U^-1
  double* const Ri = Ui + SM * SLD;      // [SM][SLD]  R~^-1
  double* const Vt = Ri + SM * SLD;      // [SRB][SLD] V of the block
  double* const Qt = Vt + SRB * SLD;     // [SRB][SLD] Q', then X~, then X, each in place
  double* const xs = Qt + SRB * SLD;     // [SRB][DT]
  double* const isr = xs + SRB * DT;     // [SRB] per-row values of the block
  double* const vr = isr + SRB;
  double* const wr = vr + SRB;
  double* const esr = wr + SRB;
  double* const q2s = esr + SRB;
  double* const qbs = q2s + SRB;
  double* const bv = qbs + SRB;          // [SM] b
  double* const tt = bv + SM;            // [SM] t~
  double* const red = tt + SM;           // [4][SM] scratch of the final column reductions
  double* const iscL = red + 4 * SM;     // MS: [SM][DT] 1 / ms_kc
  double* const es2L = iscL + SM * DT;   // MS: [SRB][DT] sum_c E_rc / ms_kc of the block's rows
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, lq = lane >> 4;
  const ExpK ek = exp_consts();
  const int d = a.d, D = a.D;
  load_corner(a.uinv, a.mp, Ui, tid);
  load_corner(a.rinv, a.mp, Ri, tid);
  if (tid < SM) {
    bv[tid] = a.bvec[tid];
    tt[tid] = a.ttil[tid];
  }
  if constexpr (MS) {
    for (int idx = tid; idx < SM * DT; idx += 256) {
      const int c = idx / DT, k = idx % DT;
      iscL[idx] = (k < d && c < a.m) ? 1.0 / a.cp.ms[(int64_t)c * d + k] : 0.0;
    }
  }
  const int col = lane, rg = wv;
  const bool live_c = col < a.m;
  // moments of E against the original inputs (`Proj derivative): per-thread sums for D <= 16; above that one more MFMA
  // product per block, X_big^T E, with the inputs staged where V was (WIDE)
  constexpr bool WIDE = DBT > 16 || MS;
  constexpr int NGB = WIDE ? 1 : DBT;
  double z[DT], gx[DT], gb[NGB];
  double isc[MS ? DT : 1], gxx[MS ? DT : 1], lsum = 0.0;
#pragma unroll
  for (int k = 0; k < DT; ++k) {
    z[k] = (k < d && live_c) ? a.Z[(int64_t)col * d + k] : 0.0;
    gx[k] = 0.0;
    if constexpr (MS) {
      const double scale = (k < d && live_c) ? a.cp.ms[(int64_t)col * d + k] : 1.0;
      isc[k] = 1.0 / scale;
      lsum += log(scale);
      gxx[k] = 0.0;
    }
  }
#pragma unroll
  for (int k = 0; k < NGB; ++k) gb[k] = 0.0;
  sd4 accGB[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) accGB[ct] = sd4{0.0, 0.0, 0.0, 0.0};
  sd4 accG[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) accG[ct] = sd4{0.0, 0.0, 0.0, 0.0};
  double cs = 0.0, sE = 0.0, sED = 0.0;
  double p_v = 0.0, p_is = 0.0, p_res = 0.0, p_v1 = 0.0;
  // `Proj second term: thread t accumulates outputs t, t + 256, ... of the D x d matrix
  constexpr int NPJ = (DBT * DT + 255) / 256;
  double pj[NPJ];
  int pj_big[NPJ], pj_small[NPJ];
#pragma unroll
  for (int j = 0; j < NPJ; ++j) {
    const int o = min(tid + 256 * j, max(D * d - 1, 0));
    pj[j] = 0.0;
    pj_big[j] = d > 0 ? o / d : 0;
    pj_small[j] = d > 0 ? o % d : 0;
  }
  const int nblk = a.rows_p / SRB;
  for (int b = blockIdx.x; b < nblk; b += gridDim.x) {
    const int r0 = b * SRB;
    __syncthreads();
    for (int idx = tid; idx < SRB * DT; idx += 256) {
      const int r = idx / DT, k = idx % DT;
      xs[idx] = (k < d && r0 + r < a.rows) ? a.pts[(int64_t)(r0 + r) * d + k] : 0.0;
    }
    load_corner(a.V + (int64_t)r0 * a.mp, a.mp, Vt, tid);
    double kreg[KR ? 16 : 1];  // this thread's K entries of the block (column, 16 rows): requested now, used in the E phase
    if constexpr (KR) {
#pragma unroll
      for (int i = 0; i < 16; ++i) kreg[i] = a.Kin[(int64_t)(r0 + rg * 16 + i) * SM + col];
    }
    if (tid < SRB) isr[tid] = a.is[r0 + tid];
    const bool rowlive = tid < SRB && r0 + tid < a.rows;  // the row phase below: one thread per row
    const double rreg = rowlive ? a.r[r0 + tid] : 0.0;
    const double yreg = (rowlive && a.y) ? a.y[r0 + tid] : 0.0;
    __syncthreads();
    sd4 acc[4];
    rows_times<false, MS ? 4 : 8>(Vt, Ri, wv, l15, lq, acc);  // Q' = V R~^-1
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double s2 = 0.0, sb = 0.0;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        s2 += acc[ct][r] * acc[ct][r];
        sb += acc[ct][r] * bv[16 * ct + l15];
      }
      s2 = sum16(s2);
      sb = sum16(sb);
      const int row = 16 * wv + lq + 4 * r;
      if (l15 == 0) {
        q2s[row] = s2;
        qbs[row] = sb;
      }
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) Qt[row * SLD + 16 * ct + l15] = acc[ct][r];
    }
    __syncthreads();
    if (tid < SRB) {  // q_diag, w, v (as pass2_rows_kernel)
      const int row = r0 + tid;
      double w = 0.0, v = 0.0, es = 0.0;
      if (row < a.rows) {
        const double is = isr[tid], rr = rreg;
        const double qd = is * q2s[tid], sb = qbs[tid];
        const double y = yreg;
        const double res = a.y ? (y - sb) : 0.0;
        w = is * res;
        const double v1 = a.variational ? is * (2.0 - is * rr - qd) : is * (1.0 - qd);
        v = v1 - w * w;
        es = qd - v * (a.cp.sf2 - rr) - w * sb;
        p_v += v;
        p_is += is;
        p_res += w * res;
        p_v1 += v1;
      }
      a.w[row] = w;
      a.v[row] = v;
      if (a.es) a.es[row] = es;
      wr[tid] = w;
      vr[tid] = v;
      esr[tid] = es;
    }
    __syncthreads();
    rows_times<true, MS ? 4 : 8>(Qt, Ri, wv, l15, lq, acc);  // Q' R~^-T
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * wv + lq + 4 * r;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const int c = 16 * ct + l15;  // X~ = diag(is) Q' R~^-T - diag(v) V - w t~^T
        Qt[row * SLD + c] = isr[row] * acc[ct][r] - vr[row] * Vt[row * SLD + c] - wr[row] * tt[c];
      }
    }
    rows_times<true, MS ? 4 : 8>(Qt, Ui, wv, l15, lq, acc);  // X = X~ U^-T
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * wv + lq + 4 * r;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        Qt[row * SLD + 16 * ct + l15] = acc[ct][r];
        if (a.X) a.X[(int64_t)(r0 + row) * a.mp + 16 * ct + l15] = acc[ct][r];
      }
    }
    gram_update(Vt, vr, wv, l15, lq, accG);  // G~ part = V^T diag(v) V
    __syncthreads();
    if constexpr (WIDE) {  // V is done with: its tile now holds the block's original inputs, zero-padded to 64 columns
      constexpr int DW = 64;
      for (int idx = tid; idx < SRB * DW; idx += 256) {
        const int r = idx / DW, k = idx % DW;
        Vt[r * SLD + k] = (k < D && r0 + r < a.rows) ? a.big[(int64_t)(r0 + r) * D + k] : 0.0;
      }
    }
    if constexpr (KR) {  // V is done with: its tile takes the block's K (rows of this wavefront)
      static_assert(!(KR && (DBT > 16 || MS)), "the staged-inputs variants need the tile themselves");
#pragma unroll
      for (int i = 0; i < 16; ++i) Vt[(rg * 16 + i) * SLD + col] = kreg[i];
    }
    // E = X .* K of the block: column sums, moments against the points (and the original inputs), sum E, sum E |x - z|^2
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
      const int r = rg * 16 + i;
      double dist = MS ? lsum : 0.0;
#pragma unroll
      for (int k = 0; k < DT; ++k) {
        const double diff = xs[r * DT + k] - z[k];
        if constexpr (MS) dist += diff * diff * isc[k];
        else dist = dist + diff * diff;
      }
      [[maybe_unused]] const bool live = live_c && r0 + r < a.rows;
      double e;
      if constexpr (KR) e = Qt[r * SLD + col] * Vt[r * SLD + col];  // (K is zero on padded rows and columns)
      else e = live ? Qt[r * SLD + col] * exp_fast(a.cp.log_sf2 + a.cp.inv_ell2_05 * dist, ek) : 0.0;
#pragma unroll
      for (int k = 0; k < DT; ++k) {
        gx[k] += xs[r * DT + k] * e;
        if constexpr (MS) gxx[k] += xs[r * DT + k] * xs[r * DT + k] * e;
      }
      if constexpr (WIDE) {
        Qt[r * SLD + col] = e;  // (rows of this wavefront)
      } else if (D > 0 && r0 + r < a.rows) {
        const double* xb = a.big + (int64_t)(r0 + r) * D;
#pragma unroll
        for (int k = 0; k < NGB; ++k)
          if (k < D) gb[k] += xb[k] * e;
      }
      cs += e;
      sE += e;
      sED += e * dist;
    }
    if constexpr (WIDE) {
      __syncthreads();  // the staged inputs and E are complete
      if constexpr (MS) {  // es2[row][k] = sum_c E_rc / ms_kc: four threads per row, sixteen columns each
        const int row = tid >> 2, part = tid & 3;
        double sum[DT];
#pragma unroll
        for (int k = 0; k < DT; ++k) sum[k] = 0.0;
#pragma unroll 2
        for (int c = 16 * part; c < 16 * part + 16; ++c) {
          const double e = Qt[row * SLD + c];
#pragma unroll
          for (int k = 0; k < DT; ++k) sum[k] += e * iscL[c * DT + k];
        }
#pragma unroll
        for (int k = 0; k < DT; ++k) {
          double t = sum[k];
          t += __shfl_xor(t, 1);
          t += __shfl_xor(t, 2);
          if (part == 0) es2L[row * DT + k] = t;
        }
      }
      // accGB[ct] += (X_big^T E) tile (wv, ct): input dimensions 16 wv .. 16 wv + 15 against columns 16 ct ..
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        double af[8], bf[8][4];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = 4 * (8 * h + j) + lq;
          af[j] = Vt[k * SLD + 16 * wv + l15];
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) bf[j][ct] = Qt[k * SLD + 16 * ct + l15];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) accGB[ct] = mfma_f64(af[j], bf[j][ct], accGB[ct]);
      }
    }
    if (D > 0) {  // second term of the `Proj derivative: sum_r x_big,r p_small,r rowsum(E)_r  (MS: E / ms_small per column)
      if constexpr (MS) __syncthreads();
      const int nr = min(SRB, a.rows - r0);
      for (int r = 0; r < nr; ++r) {
#pragma unroll
        for (int j = 0; j < NPJ; ++j) {
          const double xb = WIDE ? Vt[r * SLD + pj_big[j]] : a.big[(int64_t)(r0 + r) * D + pj_big[j]];
          const double wgt = MS ? es2L[r * DT + pj_small[j]] : esr[r];
          pj[j] += xb * xs[r * DT + pj_small[j]] * wgt;
        }
      }
    }
  }
  constexpr int MSR = MS ? 1 : 0;
  const int ncq = 1 + d + D + MSR * d;  // rows of the column block
  double* part = a.part + (int64_t)blockIdx.x * p2len(d, D, MSR);
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) part[(16 * wv + lq + 4 * r) * SM + 16 * ct + l15] = accG[ct][r];
  double* pcol = part + SM * SM;
  // per-column accumulators: the four row groups of a column are combined in order
  if constexpr (WIDE) {  // these sums are complete (the MFMA product ran over all 64 rows of every block)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = 16 * wv + lq + 4 * r;
        if (k < D) pcol[(1 + d + k) * SM + 16 * ct + l15] = accGB[ct][r];
      }
  }
  for (int q = 0; q < ncq; ++q) {
    if (WIDE && q > d && q <= d + D) continue;  // (written above)
    double val = cs;
    if (q >= 1 && q <= d) {
#pragma unroll
      for (int k = 0; k < DT; ++k)
        if (k == q - 1) val = gx[k];
    } else if (q > d + D) {
      if constexpr (MS) {
#pragma unroll
        for (int k = 0; k < DT; ++k)
          if (k == q - 1 - d - D) val = gxx[k];
      }
    } else if (q > d) {
#pragma unroll
      for (int k = 0; k < NGB; ++k)
        if (k == q - 1 - d) val = gb[k];
    }
    __syncthreads();
    red[rg * SM + col] = val;
    __syncthreads();
    if (tid < SM) pcol[q * SM + tid] = (red[tid] + red[SM + tid]) + (red[2 * SM + tid] + red[3 * SM + tid]);
  }
  double* pproj = pcol + ncq * SM;
#pragma unroll
  for (int j = 0; j < NPJ; ++j)
    if (tid + 256 * j < D * d) pproj[tid + 256 * j] = pj[j];
  double* ptail = pproj + D * d;
  sE = sum64(sE);
  sED = sum64(sED);
  __syncthreads();
  if (lane == 0) {
    red[wv] = sE;
    red[4 + wv] = sED;
  }
  __syncthreads();
  if (wv == 0) {
    p_v = sum64(p_v);
    p_is = sum64(p_is);
    p_res = sum64(p_res);
    p_v1 = sum64(p_v1);
    if (lane == 0) {
      ptail[0] = p_v;
      ptail[1] = p_is;
      ptail[2] = p_res;
      ptail[3] = p_v1;
      ptail[4] = (red[0] + red[1]) + (red[2] + red[3]);
      ptail[5] = (red[4] + red[5]) + (red[6] + red[7]);
      ptail[6] = 0.0;
      ptail[7] = 0.0;
    }
  }
}

// exchange-2 buffer from the pass-2 partials, every entry written: the (0,0) tile (zero outside its 64 x 64 corner), the
// column block (col_rows x mp; rows 0..d+D, columns < 64 carry sums), the `Proj second term and the scalar tail
__global__ __launch_bounds__(256) void small_reduce2_kernel(const double* __restrict__ part, int ng, int mp, int d, int D,
                                                            int ms, int col_rows, double* __restrict__ tile,
                                                            double* __restrict__ colblk, double* __restrict__ proj,
                                                            double* __restrict__ tail) {
  const int plen = p2len(d, D, ms), ncq = 1 + d + D + (ms ? d : 0);
  int idx = blockIdx.x * 256 + threadIdx.x;
  const int ntile = TILE * TILE, ncol = col_rows * mp, nproj = D * d;
  int src = -1;
  double* dst;
  if (idx < ntile) {
    const int r = idx / TILE, c = idx % TILE;
    dst = tile + idx;
    if (r < SM && c < SM) src = r * SM + c;
  } else if ((idx -= ntile) < ncol) {
    const int q = idx / mp, c = idx % mp;
    dst = colblk + idx;
    if (q < ncq && c < SM) src = SM * SM + q * SM + c;
  } else if ((idx -= ncol) < nproj) {
    dst = proj + idx;
    src = SM * SM + ncq * SM + idx;
  } else if ((idx -= nproj) < 8) {
    dst = tail + idx;
    src = SM * SM + ncq * SM + nproj + idx;
  } else {
    return;
  }
  *dst = (src >= 0) ? sum_parts(part + src, plen, ng) : 0.0;
}

// Finish stage of a small gradient evaluation in one workgroup (the m x m work of do_finish_enqueue on 64 x 64 corners):
//   B~^-1 = R~^-1 R~^-T (Utils.ichol, lib/utils.ml:110-113),  W~ = I - B~^-1 - t~ t~^T - G~,  W = U^-1 W~ U^-T
//   (lib/fitc_gp.ml:1196-1203), the trace terms of W against K_m and its derivatives (km_traces_kernel: :956-973,
//   lib/utils.ml:196-220), diag W, and the tails of both exchange buffers gathered behind the result block.
// MS: the multiscale trace terms of km_traces_ms_kernel (lib/cov_se_fat.ml:441-516)
template <int DT, bool MS>
__global__ __launch_bounds__(256) void small_finish_kernel(SmallFinishArgs a) {
  extern __shared__ __attribute__((aligned(16))) double small_lds[];
  double* const Ui = small_lds;        // [SM][SLD] U^-1
  double* const Ri = Ui + SM * SLD;    // [SM][SLD] R~^-1
  double* const Wt = Ri + SM * SLD;    // [SM][SLD] W~, then W
  double* const Yt = Wt + SM * SLD;    // [SM][SLD] W~ U^-T
  double* const zs = Yt + SM * SLD;    // [SM][DT]
  double* const tt = zs + SM * DT;     // [SM]
  double* const red = tt + SM;         // [4][SM]
  double* const msL = red + 4 * SM;    // MS: [SM][DT] multiscales (padding 1)
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, lq = lane >> 4;
  const int d = a.d, m = a.m;
  load_corner(a.uinv, a.mp, Ui, tid);
  load_corner(a.rinv, a.mp, Ri, tid);
  if (tid < SM) tt[tid] = a.ttil[tid];
  for (int idx = tid; idx < SM * DT; idx += 256) {
    const int c = idx / DT, k = idx % DT;
    zs[idx] = (k < d && c < m) ? a.Z[(int64_t)c * d + k] : 0.0;
    if constexpr (MS) msL[idx] = (k < d && c < m) ? a.ms[(int64_t)c * d + k] : 1.0;
  }
  for (int64_t i = tid; i < a.n_gather; i += 256) a.ex[i] = a.gather_from[i];
  double kreg[16];  // K_m entries of the trace phase below (thread = column, group of 16 rows): loaded now, used at the end
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = (tid >> 6) * 16 + i;
    kreg[i] = (r < m && lane < m) ? a.km[(int64_t)r * a.mp + lane] : 0.0;
  }
  __syncthreads();
  sd4 acc[4];
  rows_times<true>(Ri, Ri, wv, l15, lq, acc);  // B~^-1
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 16 * wv + lq + 4 * r;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const int c = 16 * ct + l15;
      const int rr = min(row, c), cc = max(row, c);  // G~ is valid in the upper triangle: mirrored, as build_w_kernel
      Wt[row * SLD + c] = (row == c ? 1.0 : 0.0) - acc[ct][r] - tt[row] * tt[c] - a.g[rr * TILE + cc];
    }
  }
  rows_times<true>(Wt, Ui, wv, l15, lq, acc);  // Y = W~ U^-T (rows of this wavefront)
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) Yt[(16 * wv + lq + 4 * r) * SLD + 16 * ct + l15] = acc[ct][r];
  __syncthreads();
  rows_times<false>(Ui, Yt, wv, l15, lq, acc);  // W = U^-1 Y
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 16 * wv + lq + 4 * r;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      Wt[row * SLD + 16 * ct + l15] = acc[ct][r];
      a.wmat[(int64_t)row * a.mp + 16 * ct + l15] = acc[ct][r];
    }
  }
  __syncthreads();
  const int col = lane, rg = wv;
  double g[DT], gm[MS ? DT : 1], s0 = 0.0, s1 = 0.0;
#pragma unroll
  for (int k = 0; k < DT; ++k) g[k] = 0.0;
  if constexpr (MS) {
#pragma unroll
    for (int k = 0; k < DT; ++k) gm[k] = 0.0;
  }
  if (col < m) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int r = rg * 16 + i;
      const double wk = Wt[r * SLD + col] * kreg[i];  // (0 beyond the real rows)
      s0 += wk;
      if constexpr (MS) {
        if (r != col) {
#pragma unroll
          for (int k = 0; k < DT; ++k) {
            if (k < d) {
              const double iscale = 1.0 / ((msL[r * DT + k] + msL[col * DT + k]) - 1.0);
              const double sdiff = (zs[r * DT + k] - zs[col * DT + k]) * iscale;
              g[k] += wk * sdiff;
              gm[k] += wk * (iscale - sdiff * sdiff);
            }
          }
        }
      } else {
        double dist = 0.0;
#pragma unroll
        for (int k = 0; k < DT; ++k) {
          const double df = zs[r * DT + k] - zs[col * DT + k];
          dist += df * df;
          g[k] += wk * df;
        }
        s1 += wk * dist;
      }
    }
  }
  for (int q = 0; q < a.km_rows; ++q) {
    double val = 0.0;
    if (q == 0) val = s0;
    else if (q == 1) val = s1;
    else if (q < 2 + d) {
#pragma unroll
      for (int k = 0; k < DT; ++k)
        if (k == q - 2) val = g[k];
    } else if constexpr (MS) {
#pragma unroll
      for (int k = 0; k < DT; ++k)
        if (k == q - 2 - d) val = gm[k];
    }
    __syncthreads();
    red[rg * SM + col] = val;
    __syncthreads();
    if (tid < SM) a.kmred[(int64_t)q * a.mp + tid] = (red[tid] + red[SM + tid]) + (red[2 * SM + tid] + red[3 * SM + tid]);
  }
  if (a.wdiag && tid < SM) a.wdiag[tid] = Wt[tid * SLD + tid];
}

// Means.calc / Variances.calc (lib/fitc_gp.ml:418-425, :498-518) for a block of 64 test points in one kernel: K tile,
// mean = K t, V = K U^-1, Q = V R~^-1, var = (sf2 - (|V_i|^2 - |Q_i|^2)) + add -- what do_predict otherwise does with seven
// launches per chunk (covariance, two triangular products, three row kernels, the combination).
template <int DT>
__global__ __launch_bounds__(256) void small_predict_kernel(SmallPredictArgs a) {
  extern __shared__ __attribute__((aligned(16))) double small_lds[];
  double* const Ui = small_lds;         // [SM][SLD]
  double* const Ri = Ui + SM * SLD;     // [SM][SLD]
  double* const Kt = Ri + SM * SLD;     // [SRB][SLD] K, then V in place
  double* const xs = Kt + SRB * SLD;    // [SRB][DT]
  double* const tv = xs + SRB * DT;     // [SM] mean coefficients
  double* const rk = tv + SM;           // [SRB] |V_i|^2
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, lq = lane >> 4;
  const ExpK ek = exp_consts();
  if (a.vars) {
    load_corner(a.uinv, a.mp, Ui, tid);
    load_corner(a.rinv, a.mp, Ri, tid);
  }
  if (tid < SM) tv[tid] = tid < a.m ? a.tvec[tid] : 0.0;
  const int col = lane, rg = wv;
  const bool live_c = col < a.m;
  double z[DT];
#pragma unroll
  for (int k = 0; k < DT; ++k) z[k] = (k < a.d && live_c) ? a.Z[(int64_t)col * a.d + k] : 0.0;
  const int r0 = blockIdx.x * SRB;
  for (int idx = tid; idx < SRB * DT; idx += 256) {
    const int r = idx / DT, k = idx % DT;
    xs[idx] = (k < a.d && r0 + r < a.rows) ? a.pts[(int64_t)(r0 + r) * a.d + k] : 0.0;
  }
  __syncthreads();
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int r = rg * 16 + i;
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < DT; ++k) {
      const double diff = xs[r * DT + k] - z[k];
      acc = acc + diff * diff;
    }
    Kt[r * SLD + col] = (r0 + r < a.rows && live_c) ? exp_fast(a.cp.log_sf2 + a.cp.inv_ell2_05 * acc, ek) : 0.0;
  }
  __syncthreads();
  if (a.means) {  // four threads per row, sixteen columns each
    const int row = tid >> 2, part = tid & 3;
    double sum = 0.0;
#pragma unroll
    for (int c = 16 * part; c < 16 * part + 16; ++c) sum += Kt[row * SLD + c] * tv[c];
    sum += __shfl_xor(sum, 1);
    sum += __shfl_xor(sum, 2);
    if (part == 0 && r0 + row < a.rows) a.means[r0 + row] = sum;
  }
  if (!a.vars) return;
  __syncthreads();  // (the mean phase read rows of other wavefronts)
  sd4 acc[4];
  rows_times<false>(Kt, Ui, wv, l15, lq, acc);  // V = K U^-1
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    double s2 = 0.0;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) s2 += acc[ct][r] * acc[ct][r];
    s2 = sum16(s2);
    const int row = 16 * wv + lq + 4 * r;
    if (l15 == 0) rk[row] = s2;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) Kt[row * SLD + 16 * ct + l15] = acc[ct][r];  // rows of this wavefront: in place
  }
  rows_times<false>(Kt, Ri, wv, l15, lq, acc);  // Q = V R~^-1
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    double s2 = 0.0;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) s2 += acc[ct][r] * acc[ct][r];
    s2 = sum16(s2);
    const int row = 16 * wv + lq + 4 * r;
    // prior_variance -. (k -. b), lib/fitc_gp.ml:475 (as variance_combine_kernel)
    if (l15 == 0 && r0 + row < a.rows) a.vars[r0 + row] = (a.cp.sf2 - (rk[row] - s2)) + a.add;
  }
}

static size_t small_lds4(int DT) { return (size_t)(3 * SM * SLD + SRB * DT + SM + SRB) * sizeof(double); }

static size_t small_lds1(int DT) { return (size_t)(2 * SM * SLD + SRB * DT + 3 * SRB) * sizeof(double); }
static size_t small_lds3(int DT, bool ms = false) {
  return (size_t)(4 * SM * SLD + SM * DT + SM + 4 * SM + (ms ? SM * DT : 0)) * sizeof(double);
}
static size_t small_lds2(int DT, bool ms = false) {
  return (size_t)(4 * SM * SLD + SRB * DT + 6 * SRB + 2 * SM + 4 * SM + (ms ? 2 * SM * DT : 0)) * sizeof(double);
}

template <typename F>
static void small_dispatch(int d, F&& go) {
  if (d <= 4) go(std::integral_constant<int, 4>{});
  else if (d <= 8) go(std::integral_constant<int, 8>{});
  else go(std::integral_constant<int, 16>{});
}

static void small_attrs() {
  static uint64_t done = 0;
  once_per_device(done, [] {
    auto set = [](const void* f, size_t bytes) {
      GPR_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    };
#define GPRHIP_SMALL_SET(DT)                                                                                   \
  set(reinterpret_cast<const void*>(&small_pass1_kernel<DT, false>), small_lds1(DT));                          \
  set(reinterpret_cast<const void*>(&small_pass2_kernel<DT, 1, false>), small_lds2(DT));                       \
  set(reinterpret_cast<const void*>(&small_pass2_kernel<DT, 16, false>), small_lds2(DT));                      \
  set(reinterpret_cast<const void*>(&small_pass2_kernel<DT, 64, false>), small_lds2(DT));                      \
  set(reinterpret_cast<const void*>(&small_pass2_kernel<(DT > 8 ? 8 : DT), 1, false, true>), small_lds2(DT > 8 ? 8 : DT));  \
  set(reinterpret_cast<const void*>(&small_pass2_kernel<(DT > 8 ? 8 : DT), 16, false, true>), small_lds2(DT > 8 ? 8 : DT)); \
  set(reinterpret_cast<const void*>(&small_finish_kernel<DT, false>), small_lds3(DT));
    GPRHIP_SMALL_SET(4)
    GPRHIP_SMALL_SET(8)
    GPRHIP_SMALL_SET(16)
#undef GPRHIP_SMALL_SET
    set(reinterpret_cast<const void*>(&small_predict_kernel<4>), small_lds4(4));
    set(reinterpret_cast<const void*>(&small_predict_kernel<8>), small_lds4(8));
    set(reinterpret_cast<const void*>(&small_predict_kernel<16>), small_lds4(16));
#define GPRHIP_SMALL_SET_MS(DT)                                                                                \
  set(reinterpret_cast<const void*>(&small_pass1_kernel<DT, true>), small_lds1(DT));                           \
  set(reinterpret_cast<const void*>(&small_pass2_kernel<DT, 1, true>), small_lds2(DT, true));                  \
  set(reinterpret_cast<const void*>(&small_pass2_kernel<DT, 64, true>), small_lds2(DT, true));                 \
  set(reinterpret_cast<const void*>(&small_finish_kernel<DT, true>), small_lds3(DT, true));
    GPRHIP_SMALL_SET_MS(4)
    GPRHIP_SMALL_SET_MS(8)
#undef GPRHIP_SMALL_SET_MS
  });
}

bool small_path_fits(int m, int mp, int d, int D, int64_t rows, bool ms) {
  // (rows: no structural limit -- the partial sums are per workgroup, not per block; 4M rows is where int indices of the
  //  row kernels around it were last checked.  Multiscales: d <= 8, their extra LDS arrays do not fit beside d = 16.)
  return m <= SM && mp == TILE && d <= (ms ? 8 : 16) && D <= 64 && rows <= (int64_t(1) << 22);
}

void launch_small_pass1(const SmallPass1Args& a, double* tile, double* cvec, double* tail, hipStream_t s) {
  small_attrs();
  const int ng = small_groups(a.rows_p, SMALL_GROUPS1);
  const bool ms = a.cp.ms != nullptr;
  small_dispatch(a.d, [&](auto dt) {
    constexpr int DT = decltype(dt)::value;
    if constexpr (DT <= 8) {
      if (ms) {
        hipLaunchKernelGGL((small_pass1_kernel<DT, true>), dim3(ng), dim3(256), small_lds1(DT), s, a);
        return;
      }
    }
    hipLaunchKernelGGL((small_pass1_kernel<DT, false>), dim3(ng), dim3(256), small_lds1(DT), s, a);
  });
  hipLaunchKernelGGL(small_reduce1_kernel, dim3(TILE * TILE / 256 + 1), dim3(256), 0, s, a.part, ng, a.mp, tile, cvec, tail);
  GPR_HIP(hipGetLastError());
}

void launch_small_pass2(const SmallPass2Args& a, int col_rows, double* tile, double* colblk, double* proj, double* tail,
                        hipStream_t s) {
  small_attrs();
  const int ng = small_groups(a.rows_p, SMALL_GROUPS2);
  const bool ms = a.cp.ms != nullptr;
  small_dispatch(a.d, [&](auto dt) {
    constexpr int DT = decltype(dt)::value;
    if constexpr (DT <= 8) {
      if (ms) {
        if (a.D == 0) hipLaunchKernelGGL((small_pass2_kernel<DT, 1, true>), dim3(ng), dim3(256), small_lds2(DT, true), s, a);
        else hipLaunchKernelGGL((small_pass2_kernel<DT, 64, true>), dim3(ng), dim3(256), small_lds2(DT, true), s, a);
        return;
      }
    }
    if constexpr (DT <= 8) {
      if (a.Kin && a.D <= 16) {
        if (a.D == 0) hipLaunchKernelGGL((small_pass2_kernel<DT, 1, false, true>), dim3(ng), dim3(256), small_lds2(DT), s, a);
        else hipLaunchKernelGGL((small_pass2_kernel<DT, 16, false, true>), dim3(ng), dim3(256), small_lds2(DT), s, a);
        return;
      }
    }
    if (a.D == 0) hipLaunchKernelGGL((small_pass2_kernel<DT, 1, false>), dim3(ng), dim3(256), small_lds2(DT), s, a);
    else if (a.D <= 16) hipLaunchKernelGGL((small_pass2_kernel<DT, 16, false>), dim3(ng), dim3(256), small_lds2(DT), s, a);
    else hipLaunchKernelGGL((small_pass2_kernel<DT, 64, false>), dim3(ng), dim3(256), small_lds2(DT), s, a);
  });
  const int nout = TILE * TILE + col_rows * a.mp + a.D * a.d + 8;
  hipLaunchKernelGGL(small_reduce2_kernel, dim3((nout + 255) / 256), dim3(256), 0, s, a.part, ng, a.mp, a.d, a.D, ms ? 1 : 0,
                     col_rows, tile, colblk, proj, tail);
  GPR_HIP(hipGetLastError());
}

void launch_small_predict(const SmallPredictArgs& a, hipStream_t s) {
  small_attrs();
  small_dispatch(a.d, [&](auto dt) {
    constexpr int DT = decltype(dt)::value;
    hipLaunchKernelGGL((small_predict_kernel<DT>), dim3((a.rows + SRB - 1) / SRB), dim3(256), small_lds4(DT), s, a);
  });
  GPR_HIP(hipGetLastError());
}

void launch_small_finish(const SmallFinishArgs& a, hipStream_t s) {
  small_attrs();
  small_dispatch(a.d, [&](auto dt) {
    constexpr int DT = decltype(dt)::value;
    if constexpr (DT <= 8) {
      if (a.ms) {
        hipLaunchKernelGGL((small_finish_kernel<DT, true>), dim3(1), dim3(256), small_lds3(DT, true), s, a);
        return;
      }
    }
    hipLaunchKernelGGL((small_finish_kernel<DT, false>), dim3(1), dim3(256), small_lds3(DT), s, a);
  });
  GPR_HIP(hipGetLastError());
}

}  // namespace gprhip
