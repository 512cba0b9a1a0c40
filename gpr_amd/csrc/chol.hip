// Diagonal-block kernels of the blocked upper Cholesky (dpotrf `U) and triangular inverse,
// plus the small m-vector helpers.  The trailing updates and panel solves run on the MFMA
// engine; these kernels handle the 128 x 128 diagonal blocks in LDS (one workgroup each).
// Reference call sites: lib/fitc_gp.ml:53-57 (potrf of K_m + jitter), lib/utils.ml:95-113.
#include <cstdlib>
#include <memory>
#include <vector>
#include "kernels.h"
#include "exp_fast.h"

namespace gprhip {

constexpr int NB = TILE;        // 128
constexpr int LDT = NB + 1;     // LDS row stride (bank spread)
constexpr int MB = 16;          // micro-panel width
constexpr int PT = 512;         // threads of the diagonal-block kernel
// LDS: T[NB][LDT] | T1[NB][MB] (product scratch of the inversion) | rdiag[NB] | flag
constexpr int POTRF_LDS = (NB * LDT + NB * MB + NB + 3 * NB) * 8 + 16;  // + c~, b, t~ of the fused B~ phase

// A = U^T U in place on block j (upper); strict lower of the block zeroed; dinv = inv(U_jj).
//
// Both phases advance by 16-column micro-panels so that the sequential part runs inside one
// wavefront on registers (cross-lane broadcasts, no workgroup barrier):
//   factor : 16x16 diagonal block (wave 0, registers) -> 16 x rest panel solve (thread per column,
//            reciprocal pivots) -> rank-16 update of the trailing block as 16x16 MFMA tiles (all wavefronts)
//   invert : the eight 16x16 diagonal inverses first, all at once (kept in the unused strictly-lower
//            blocks of T); then per block column  T1 = X[0:j0,0:j0] U[0:j0,j],  X[0:j0,j] = -T1 inv(U_jj)
//            in place (LAPACK dtrtri order), one 16x16 MFMA tile per wavefront.
// broadcast lane `l` (compile-time constant after unrolling) through v_readlane, not the LDS crossbar
__device__ __forceinline__ double bcast_lane(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

typedef double pd4 __attribute__((ext_vector_type(4)));
// v_mfma_f64_16x16x4_f64: lane supplies A[lane&15][lane>>4] and B[lane>>4][lane&15]; accumulator element r is
// D[(lane>>4) + 4r][lane&15]
__device__ __forceinline__ pd4 mfma_f64(double a, double b, pd4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// ---- 16-pivot chain of a micro diagonal block on DPP broadcasts (round 5)
// v_mov_b64_dpp / v_fmac_f64_dpp with row_newbcast:N (gfx90a+): lane N of each 16-lane row to every lane of that row, as
// the first source of the instruction itself -- ONE VALU instruction per eliminated entry where v_readlane needs two plus
// an SGPR round trip and a separate FMA.  The four rows of a wavefront hold the same columns (lane & 15) and compute the
// same thing.  The compiler does not look inside inline assembly for hazards: a DPP source written by the VALU
// instruction just before needs two wait states, so every block starts with `s_nop 1` and holds all DPP reads of a
// pivot's row in one statement (nothing the register allocator inserts can land between them).
#define CH_F(R, X) "v_fmac_f64_dpp %[" #X #R "], %[u], -%[w] row_newbcast:" #R " row_mask:0xf bank_mask:0xf\n\t"
#define CH_R15(X) CH_F(15, X)
#define CH_R14(X) CH_F(14, X) CH_R15(X)
#define CH_R13(X) CH_F(13, X) CH_R14(X)
#define CH_R12(X) CH_F(12, X) CH_R13(X)
#define CH_R11(X) CH_F(11, X) CH_R12(X)
#define CH_R10(X) CH_F(10, X) CH_R11(X)
#define CH_R9(X) CH_F(9, X) CH_R10(X)
#define CH_R8(X) CH_F(8, X) CH_R9(X)
#define CH_R7(X) CH_F(7, X) CH_R8(X)
#define CH_R6(X) CH_F(6, X) CH_R7(X)
#define CH_R5(X) CH_F(5, X) CH_R6(X)
#define CH_R4(X) CH_F(4, X) CH_R5(X)
#define CH_R3(X) CH_F(3, X) CH_R4(X)
#define CH_R2(X) CH_F(2, X) CH_R3(X)
#define CH_R1(X) CH_F(1, X) CH_R2(X)
#define CH_OPS(v)                                                                                                       \
  [v##1] "+v"(v[1]), [v##2] "+v"(v[2]), [v##3] "+v"(v[3]), [v##4] "+v"(v[4]), [v##5] "+v"(v[5]), [v##6] "+v"(v[6]),     \
      [v##7] "+v"(v[7]), [v##8] "+v"(v[8]), [v##9] "+v"(v[9]), [v##10] "+v"(v[10]), [v##11] "+v"(v[11]),                \
      [v##12] "+v"(v[12]), [v##13] "+v"(v[13]), [v##14] "+v"(v[14]), [v##15] "+v"(v[15])
// pivot Q of the chain: broadcast, reciprocal square root, scale row Q of [A | Y], eliminate it from rows Q+1..15
// (a[r] -= U[q][r] U[q][c], y[r] -= U[q][r] Y[q][c]: the identity right-hand side rides along, so the block's inverse
// transpose comes out of the same steps on registers the factor alone would leave idle)
#define CH_PIVOT(Q, ROWS)                                                                                                \
  {                                                                                                                      \
    double dq;                                                                                                           \
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:" #Q " row_mask:0xf bank_mask:0xf" : "=v"(dq) : "v"(a[Q])); \
    /* The dependent path of a pivot is what the chain costs (a dependent fp64 operation is ~20 cycles here): broadcast, \
       v_rsq_f64, three levels of its third-order correction, the row scaling, the first elimination -- 8 levels.  The   \
       positivity test and the pivot's own square root hang off it, not in it: a non-positive pivot is recorded and the  \
       numbers behind it go NaN/Inf (the caller reads `info` and discards them). */                                      \
    badq = (!(dq > 0.0) && badq == 0) ? k0 + Q + 1 : badq;                                                               \
    double rs = __builtin_amdgcn_rsq(dq);                                                                                \
    const double e = fma(-dq * rs, rs, 1.0);                                                                             \
    rs = fma(rs * e, fma(0.375, e, 0.5), rs); /* rsqrt to ~1 ulp */                                                      \
    const double uqc = a[Q] * rs;                                                                                        \
    const double yq = y[Q] * rs;                                                                                         \
    ROWS                                                                                                                 \
    double piv = dq * rs;                                                                                                \
    piv = fma(fma(-piv, piv, dq), 0.5 * rs, piv); /* sqrt(dq), one Heron correction */                                   \
    a[Q] = (cc == Q) ? piv : uqc;                                                                                        \
    y[Q] = yq;                                                                                                           \
    myrp = (cc == Q) ? rs : myrp;                                                                                        \
  }
#define CH_ELIM(RM)                                                                    \
  asm volatile("s_nop 1\n\t" RM(a) : CH_OPS(a) : [u] "v"(uqc), [w] "v"(uqc));           \
  asm volatile("s_nop 1\n\t" RM(y) : CH_OPS(y) : [u] "v"(uqc), [w] "v"(yq));
// a[r] = column cc of the 16 x 16 block (row r), y[r] = column cc of the identity; on return a = U (rows <= cc valid),
// y = column cc of U^-T, myrp = 1 / U[cc][cc], badq = first non-positive pivot (1-based, + k0) or unchanged
__device__ __forceinline__ void chain16(double (&a)[16], double (&y)[16], int cc, int k0, int& badq, double& myrp) {
  CH_PIVOT(0, CH_ELIM(CH_R1))
  CH_PIVOT(1, CH_ELIM(CH_R2))
  CH_PIVOT(2, CH_ELIM(CH_R3))
  CH_PIVOT(3, CH_ELIM(CH_R4))
  CH_PIVOT(4, CH_ELIM(CH_R5))
  CH_PIVOT(5, CH_ELIM(CH_R6))
  CH_PIVOT(6, CH_ELIM(CH_R7))
  CH_PIVOT(7, CH_ELIM(CH_R8))
  CH_PIVOT(8, CH_ELIM(CH_R9))
  CH_PIVOT(9, CH_ELIM(CH_R10))
  CH_PIVOT(10, CH_ELIM(CH_R11))
  CH_PIVOT(11, CH_ELIM(CH_R12))
  CH_PIVOT(12, CH_ELIM(CH_R13))
  CH_PIVOT(13, CH_ELIM(CH_R14))
  CH_PIVOT(14, CH_ELIM(CH_R15))
  CH_PIVOT(15, )
}

__device__ __forceinline__ double* dblk(double* T, int b) {
  // home of the inverse of diagonal micro-block b: the free block just below the diagonal (b<7: (b+1,b); 7: (7,0))
  return (b < 7) ? T + ((b + 1) * MB) * LDT + b * MB : T + (7 * MB) * LDT;
}

// m_real (0 = mp): rows / columns at and beyond it are identity padding (K_m + jitter and B~ are 1 on the padded
// diagonal, 0 off it): the 16-column micro-panels that lie wholly in the padding are skipped in both phases -- their
// factor and their inverse are the identity that is already there.  (m = 50: four of eight micro-panels, 47 -> 27 us.)
// FUSED (single-block matrices, mp = 128, PotrfFuse in kernels.h): the B~ phase of pass 2 in one kernel -- the block is
// I + the reduced accumulation of pass 1 instead of a load, and the m-vectors that follow the factorisation (log|B~|, b,
// t~, t, |b|^2) are formed from the inverse while it is still in LDS: six launches of ~5 us each on the latency chain of
// every evaluation with m <= 128 become the head and tail of this one.  (Measured and dropped: summing the small row
// pass's per-workgroup partials here instead of in their own launch: 1 MB through one CU, 130 us.)
// MODE 0: the block is loaded; 1 (FUSED): the B~ phase; 2: K_m + jitter of at most 64 inducing points built in place of the
// load (PotrfKm in kernels.h: the values of cov_upper_kernel from the points staged in LDS -- 4096 entries, eight per
// thread -- and the plain covariance written to km), which takes the cov_upper launch off small evaluations.
// profiling aid of tools/potrf_check (flags bit 8): s_memtime stamps of the factor's phases, thread 0
__device__ unsigned long long g_potrf_ts[64];
// Factor of the 128 x 128 block held in LDS (T[NB][LDT], upper triangle valid): A = U^T U in place, reciprocal pivots in
// rdiag, the eight 16 x 16 micro inverses D_b = inv(U_bb) at dblk(T, b), the first non-positive pivot (1-based) in *bad.
// k_end: columns that hold real rows (a multiple of 16; the rest is identity padding and is skipped); carry: an identity
// right-hand side in T[r][64 + c] (r, c < 64) goes through the same steps and ends as U^-T (blocks of at most 64 real rows).
// All PT threads call it; it ends behind a barrier.
__device__ __forceinline__ void potrf_factor_lds(double* __restrict__ T, double* __restrict__ rdiag, int* bad, int k_end,
                                                 bool carry, bool ts) {
  // Round 5.  Wavefront 0 does nothing but the chain: per 16-column micro-panel k
  //     chain16 on block (k, k) held in registers -> U_kk, reciprocal pivots, D_k = inv(U_kk) (to dblk(T, k))
  //     -- barrier 1 --
  //     X_{k+1} = D_k^T T[k, k+1]  (one 16 x 16 x 16 MFMA product), stored for the others
  //     -- barrier 2 --
  //     block (k+1, k+1) -= X_{k+1}^T X_{k+1}: four MFMAs whose A and B operands are both the accumulator registers of
  //     X_{k+1} as they stand (lane (c, q) register r holds X[q + 4r][c], which is A[c][k] and B[k][c] of k-step r), then
  //     through LDS into the chain's column layout
  // while wavefronts 1..7 solve the other tiles of the panel between the two barriers (X_j = D_k^T T[k, j], and the
  // non-zero tiles of a carried right-hand side) and apply the whole rest of the rank-16 update behind barrier 2 -- under
  // the next chain, which is the longer of the two.  Measured per micro-panel (s_memtime, tools/potrf_check TS=1): the
  // three-phase version of the same round (chain | panel | row update, every wavefront in every phase) spent 7500 cycles
  // of which the chain is 3700 (tools/chain_lab); rounds 3-4 (v_readlane chain, thread-per-column substitution): 10 200.
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  constexpr int YC = 64;  // first column of the carried right-hand side
  // helper wavefronts: all but wavefront 0 and its SIMD partner (wavefront i runs on SIMD i % 4: wavefront 4 would share
  // the chain's issue port -- with it busy the chain took 4500 - 5400 cycles per micro-panel instead of 3700)
  constexpr int NH = PT / 64 - 2;
  const int nmp = k_end / MB;
  // tile (ci, cj) -= X_ci^T X_cj with X = rows k0p.. of T (the solved panel)
  auto update_tile = [&](int k0p, int ci, int cj) {
    double* Ct = T + (ci + lq) * LDT + cj + l15;
    pd4 acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = Ct[4 * r * LDT];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const double* urow = T + (k0p + 4 * kk + lq) * LDT;
      acc = mfma_f64(-urow[ci + l15], urow[cj + l15], acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) Ct[4 * r * LDT] = acc[r];
  };
  // padded micro-blocks are their own inverse
  for (int b = nmp + wid; b < NB / MB; b += PT / 64) {
    double* D = dblk(T, b);
#pragma unroll
    for (int r = 0; r < 4; ++r) D[(lq + 4 * r) * LDT + l15] = (lq + 4 * r == l15) ? 1.0 : 0.0;
  }
  if (ts && tid == 0) g_potrf_ts[0] = clock64();
  if (wid == 0) {
    const int cc = l15;
    double a[MB], y[MB];
#pragma unroll
    for (int r = 0; r < MB; ++r) a[r] = T[r * LDT + cc];
    for (int k = 0; k < nmp; ++k) {
      const int k0 = k * MB, k1 = k0 + MB;
#pragma unroll
      for (int r = 0; r < MB; ++r) y[r] = (r == cc) ? 1.0 : 0.0;
      int badq = 0;
      double myrp = 0.0;
      chain16(a, y, cc, k0, badq, myrp);
      if (lane < MB) {
        rdiag[k0 + cc] = myrp;
#pragma unroll
        for (int r = 0; r < MB; ++r)
          if (r <= cc) T[(k0 + r) * LDT + k0 + cc] = a[r];
      } else if (lane < 2 * MB) {  // the second row of lanes holds the same numbers: it stores D_k = Y^T (row cc of D_k)
        double* D = dblk(T, k);
#pragma unroll
        for (int r = 0; r < MB; ++r) D[cc * LDT + r] = y[r];
      }
      if (lane == 0 && badq != 0 && *bad == 0) *bad = badq;
      if (ts && lane == 0) g_potrf_ts[1 + 4 * k] = clock64();
      __syncthreads();  // 1: D_k is there; the others are through with micro-panel k - 1
      if (ts && lane == 0) g_potrf_ts[2 + 4 * k] = clock64();
      pd4 x = {0.0, 0.0, 0.0, 0.0};
      if (k + 1 < nmp) {
        const double* D = dblk(T, k);
        double* Tp = T + (k0 + lq) * LDT + k1 + l15;
        double bfr[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) bfr[kk] = Tp[4 * kk * LDT];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) x = mfma_f64(D[(4 * kk + lq) * LDT + l15], bfr[kk], x);
#pragma unroll
        for (int r = 0; r < 4; ++r) Tp[4 * r * LDT] = x[r];
      }
      if (ts && lane == 0) g_potrf_ts[3 + 4 * k] = clock64();
      __syncthreads();  // 2: the whole panel is solved
      if (k + 1 < nmp) {
        double* Ct = T + (k1 + lq) * LDT + k1 + l15;
        pd4 acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = Ct[4 * r * LDT];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = mfma_f64(-x[r], x[r], acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) Ct[4 * r * LDT] = acc[r];
        // (same wavefront: the reads below are ordered behind these writes by the LDS queue itself)
#pragma unroll
        for (int r = 0; r < MB; ++r) a[r] = T[(k1 + r) * LDT + k1 + cc];
      }
      if (ts && lane == 0) g_potrf_ts[4 + 4 * k] = clock64();
    }
  } else if (wid == 4) {
    for (int k = 0; k < nmp; ++k) {
      __syncthreads();  // 1
      __syncthreads();  // 2
    }
  } else {
    const int w = wid < 4 ? wid - 1 : wid - 2;
    for (int k = 0; k < nmp; ++k) {
      const int k0 = k * MB;
      __syncthreads();  // 1
      // the other tiles of panel k: columns right of block k + 1, and the non-zero tiles of Y in these rows
      const int np = k_end / MB - (k + 2), nyt = carry ? k + 1 : 0;
      {
        const double* D = dblk(T, k);
        for (int t = w; t < (np > 0 ? np : 0) + nyt; t += NH) {
          const int ct = (t < np) ? (k + 2 + t) * MB : YC + (t - (np > 0 ? np : 0)) * MB;
          double* Tp = T + (k0 + lq) * LDT + ct + l15;
          pd4 x = {0.0, 0.0, 0.0, 0.0};
          double bfr[4];
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) bfr[kk] = Tp[4 * kk * LDT];
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) x = mfma_f64(D[(4 * kk + lq) * LDT + l15], bfr[kk], x);
#pragma unroll
          for (int r = 0; r < 4; ++r) Tp[4 * r * LDT] = x[r];
        }
      }
      __syncthreads();  // 2
      // rank-16 update of micro-panel k, everything but block (k + 1, k + 1): block row k + 1 first (the next panel reads
      // it), then the rows below, then -- when the inverse is carried -- the tiles of Y below the panel
      const int b1 = k + 1, nt = k_end / MB - b1;
      if (nt > 0) {
        const int nsym = nt * (nt + 1) / 2 - 1, ny = carry ? nt * b1 : 0;
        for (int t = w; t < nsym + ny; t += NH) {
          int ci, cj;
          if (t < nsym) {
            int ti = 0, rem = t + 1;  // (item 0 of the symmetric list is block (k + 1, k + 1): wavefront 0's)
            while (rem >= nt - ti) {
              rem -= nt - ti;
              ++ti;
            }
            ci = (b1 + ti) * MB;
            cj = (b1 + ti + rem) * MB;
          } else {
            const int e = t - nsym;
            ci = (b1 + e / b1) * MB;
            cj = YC + (e % b1) * MB;
          }
          update_tile(k0, ci, cj);
        }
      }
    }
  }
  __syncthreads();
}


template <int MODE>
__device__ __forceinline__ void potrf_diag_body(double* __restrict__ A, int mp, int j, double* __restrict__ dinv,
                                                int* __restrict__ info, int flags, int m_real, const PotrfFuse& f,
                                                const PotrfKm& g) {
  constexpr bool FUSED = MODE == 1;
  extern __shared__ __attribute__((aligned(16))) double T[];  // [NB][LDT]
  double* T1 = T + NB * LDT;                                   // [NB][MB]
  double* rdiag = T1 + NB * MB;                                // [NB] reciprocal pivots
  double* cv = rdiag + NB;                                     // [NB] c~, [NB] b, [NB] t~ (FUSED mode 2)
  int& bad = *reinterpret_cast<int*>(cv + 3 * NB);             // keep all LDS in the one dynamic array
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  j += blockIdx.x;  // a launch over several blocks handles block j + blockIdx.x (inverse-only pass over all blocks)
  dinv += (int64_t)blockIdx.x * NB * NB;
  const int live = (m_real > 0) ? min(NB, max(1, m_real - j * NB)) : NB;  // real rows of this block
  const int k_end = (live + MB - 1) / MB * MB;                              // micro-panels that hold any of them
  double* Ab = A + (int64_t)j * NB * mp + (int64_t)j * NB;
  // FUSED: thread (i = tid & 127, q = tid >> 7) keeps U^-1[i][32 q .. 32 q + 31] for t = U^-1 t~ at the end
  double ureg[NB / (PT / NB)];
  if (FUSED) {
    // B~ = I + the reduced accumulation of pass 1 ((0,0) tile of the exchange-1 buffer)
    for (int idx = tid; idx < NB * NB / 2; idx += PT) {
      const int r = idx / (NB / 2), c2 = (idx % (NB / 2)) * 2;
      if (c2 + 1 >= r) {
        const double2 v = *reinterpret_cast<const double2*>(f.src + (int64_t)r * NB + c2);
        T[r * LDT + c2] = v.x + (c2 == r ? 1.0 : 0.0);
        T[r * LDT + c2 + 1] = v.y + (c2 + 1 == r ? 1.0 : 0.0);
      }
    }
    if (tid < NB) cv[tid] = f.cvec[tid];
    else if (tid < NB + 4 && f.tail_out) f.tail_out[tid - NB] = f.tail_in[tid - NB];
    // issued now, used after the inversion: their latency hides under the factorisation (U^-1 is zero below its diagonal)
    {
      const double2* urow = reinterpret_cast<const double2*>(f.uinv + (int64_t)(tid & (NB - 1)) * NB + 32 * (tid / NB));
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        const double2 v = urow[kk];
        ureg[2 * kk] = v.x;
        ureg[2 * kk + 1] = v.y;
      }
    }
  } else if (MODE == 2) {
    // K_m (lib/cov_se_iso.ml:56-87, lib/cov_se_fat.ml:85-100) + heteroskedastic noise + jitter on the real diagonal
    // (lib/fitc_gp.ml:54-55), 1 on the padded one; g.km = the covariance alone, full and symmetric, padding 0
    const ExpK ek = exp_consts();
    double* zs = T1;  // [64][d] (d <= 16: the scratch of the inversion, free until then)
    for (int idx = tid; idx < g.m * g.d; idx += PT) zs[idx] = g.Z[idx];
    for (int idx = tid; idx < NB * NB; idx += PT) {  // everything outside the 64 x 64 corner
      const int r = idx / NB, c = idx % NB;
      if (r < 64 && c < 64) continue;
      g.km[idx] = 0.0;
      if (c >= r) T[r * LDT + c] = (r == c) ? 1.0 : 0.0;
    }
    __syncthreads();
    for (int idx = tid; idx < 64 * 64; idx += PT) {
      const int r = idx >> 6, c = idx & 63;
      double val = 0.0, valj = (r == c) ? 1.0 : 0.0;
      if (r < g.m && c < g.m) {
        if (r == c) {
          val = g.cp.sf2;
          valj = (g.het ? g.cp.sf2 + g.het[c] : g.cp.sf2) + g.jitter;
        } else {
          double acc = 0.0;
          for (int k = 0; k < g.d; ++k) {
            const double diff = zs[c * g.d + k] - zs[r * g.d + k];
            acc = acc + diff * diff;
          }
          val = exp_fast(g.cp.log_sf2 + g.cp.inv_ell2_05 * acc, ek);
          valj = val;
        }
      }
      g.km[r * NB + c] = val;
      if (c >= r) T[r * LDT + c] = valj;
    }
  } else {
    for (int idx = tid; idx < NB * NB / 2; idx += PT) {
      const int r = idx / (NB / 2), c2 = (idx % (NB / 2)) * 2;
      if (c2 + 1 >= r) {
        const double2 v = *reinterpret_cast<const double2*>(Ab + (int64_t)r * mp + c2);
        T[r * LDT + c2] = v.x;
        T[r * LDT + c2 + 1] = v.y;
      }
    }
  }
  if (tid == 0) bad = 0;
  if (tid < NB && tid >= k_end) rdiag[tid] = 1.0;  // padding: unit pivots
  // At most 64 real rows (and no ablation flags): the inverse is not formed by the pass further down but carried through
  // the factorisation as an identity right-hand side Y, kept in the block's own zero padding -- T[r][64 + c], r, c < 64,
  // is zero above the padded diagonal and takes no part in the factor.  The panel solve then also turns rows k0..k0+15
  // of Y into U11^-T Y, the trailing update subtracts U12^T Y from the rows below, and Y ends as U^-T: the steps of
  // potrf_upper_blocked on 16-column micro-panels.  They run on threads the factor leaves idle, so the inverse costs no
  // phase of its own (m = 50: 10.5 us of 32).
  const bool carry = k_end <= 64 && (flags & ~(64 | 256)) == 0;
  constexpr int YC = 64;  // first column of Y
  if (carry) {
    __syncthreads();  // (the loads above wrote zeros where Y goes)
    for (int idx = tid; idx < 64 * 64; idx += PT) T[(idx >> 6) * LDT + YC + (idx & 63)] = ((idx >> 6) == (idx & 63)) ? 1.0 : 0.0;
  }
  __syncthreads();
  if (flags & 1) {  // block already holds a factor (model import): only its inverse is wanted
    if (tid < NB) rdiag[tid] = 1.0 / T[tid * LDT + tid];
    __syncthreads();
  }

  // ---------------- factor
  if (!(flags & 1)) potrf_factor_lds(T, rdiag, &bad, k_end, carry, (flags & 256) != 0);
  else {  // (already a factor: every micro-block's inverse is formed by back substitution below)
  }
  // flags bit 6 (single-block matrices): the flag is this kernel's alone -- written either way, no memset in front of it
  if (tid == 0) {
    if (flags & 64) *info = bad ? j * NB + bad : 0;
    else if (bad != 0) atomicCAS(info, 0, j * NB + bad);
  }
  // write U back (zero strict lower of the block)
  for (int idx = tid; idx < NB * NB / 2; idx += PT) {
    const int r = idx / (NB / 2), c2 = (idx % (NB / 2)) * 2;
    double2 v;
    v.x = (c2 >= r) ? T[r * LDT + c2] : 0.0;
    v.y = (c2 + 1 >= r) ? T[r * LDT + c2 + 1] : 0.0;
    if (carry && r < 64 && c2 >= YC) v.x = v.y = 0.0;  // (Y lives there)
    *reinterpret_cast<double2*>(Ab + (int64_t)r * mp + c2) = v;
  }
  __syncthreads();
  if (carry) {
    // X = Y^T onto the upper triangle (the factor has gone to memory), then the padding is cleared again
    for (int idx = tid; idx < 64 * 64; idx += PT) {
      const int r = idx >> 6, c = idx & 63;
      if (c >= r) T[r * LDT + c] = T[c * LDT + YC + r];
    }
    __syncthreads();
    for (int idx = tid; idx < 64 * 64; idx += PT) T[(idx >> 6) * LDT + YC + (idx & 63)] = 0.0;
    __syncthreads();
  }

  // ---------------- invert in place (upper)
  if ((!(flags & 2) || (flags & 32)) && !carry) {
    // the eight diagonal micro-block inverses come out of the factor's pivot chains; a block that arrived already
    // factored (flags bit 0) gets them by back substitution here: wave w handles block w (8 waves)
    if (flags & 1) {
      const int b = wid, j0 = b * MB, cc = lane & 15;
      double x[MB];
#pragma unroll
      for (int r = MB - 1; r >= 0; --r) {
        double s = 0.0;
#pragma unroll
        for (int k = r + 1; k < MB; ++k) {
          const double u = T[(j0 + r) * LDT + j0 + k];  // same address in every lane: broadcast
          if (k <= cc) s += u * x[k];
        }
        const double rd = rdiag[j0 + r];
        x[r] = (r == cc) ? rd : ((r < cc) ? -s * rd : 0.0);
      }
      double* D = dblk(T, b);
      if (lane < MB) {
#pragma unroll
        for (int r = 0; r < MB; ++r) D[r * LDT + cc] = x[r];
      }
    }
    __syncthreads();
    if (flags & 32) {
      // factor-only step of the blocked factorisation (potrf_upper_blocked): the panel solve works by substitution
      // with the eight 16 x 16 micro inverses, stored as [8][16][16] at the head of this block's dinv slot; the
      // full block inverse is formed later, off the critical path, by the inverse-only pass (flags = 1)
      for (int idx = tid; idx < 8 * MB * MB; idx += PT)
        dinv[idx] = dblk(T, idx >> 8)[((idx >> 4) & 15) * LDT + (idx & 15)];
      return;
    }
  }
  if (!(flags & 2) && !carry) {
    // X = U^-1 in place, all eight block columns at once (round 5; the dtrtri order of rounds 1-4 walked them one after
    // the other, two barriers each: 12 us).  Wavefront c owns block column c and solves it upwards by back substitution on
    // the micro inverses:  X_cc = D_c,   X_bc = -D_b sum_{b < k <= c} U_bk X_kc   (b = c - 1 .. 0),
    // its tiles held in registers -- an accumulator tile is the B operand of the next product as it stands (rows q + 4r in
    // register r), the A operands (U_bk, D_b from LDS) are indexed to match.  The longest column is 36 tile products
    // (3.8 us of MFMAs); U is overwritten only after every wavefront has read what it needs.
    const int l15 = lane & 15, lq = lane >> 4;
    const int c = wid, nblk = k_end / MB;
    pd4 xs[8];
    if (c < nblk) {  // (block columns wholly in the padding are already their own inverse)
#pragma unroll
      for (int bb = 7; bb >= 0; --bb) {
        if (bb > c) continue;
        if (bb == c) {
          const double* D = dblk(T, bb);
#pragma unroll
          for (int r = 0; r < 4; ++r) xs[bb][r] = D[(lq + 4 * r) * LDT + l15];
          continue;
        }
        pd4 t = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 7; k > bb; --k) {
          if (k > c) continue;
          const double* Ub = T + (bb * MB + l15) * LDT + k * MB + lq;
#pragma unroll
          for (int r = 0; r < 4; ++r) t = mfma_f64(Ub[4 * r], xs[k][r], t);
        }
        const double* D = dblk(T, bb) + l15 * LDT + lq;
        pd4 x = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int r = 0; r < 4; ++r) x = mfma_f64(-D[4 * r], t[r], x);
        xs[bb] = x;
      }
    }
    __syncthreads();
    if (c < nblk) {
#pragma unroll
      for (int bb = 0; bb < 8; ++bb) {
        if (bb > c) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) T[(bb * MB + lq + 4 * r) * LDT + c * MB + l15] = xs[bb][r];
      }
    }
    __syncthreads();
  }
  if (flags & 2) return;
  for (int idx = tid; idx < NB * NB / 2; idx += PT) {
    const int r = idx / (NB / 2), c2 = (idx % (NB / 2)) * 2;
    double2 v;
    v.x = (c2 >= r) ? T[r * LDT + c2] : 0.0;
    v.y = (c2 + 1 >= r) ? T[r * LDT + c2 + 1] : 0.0;
    *reinterpret_cast<double2*>(dinv + (int64_t)r * NB + c2) = v;
  }
  if (FUSED) {
    // with X = R~^-1 (upper part of T):  b = X^T c~ (= Q_n^T y~, lib/fitc_gp.ml:285-286),  t~ = X b,  t = U^-1 t~ (:291),
    // log|B~| = -2 sum log(reciprocal pivots) (lib/utils.ml:95-101),  |b|^2
    double* bv = cv + NB;
    double* tv = bv + NB;
    // four threads per entry, a quarter of the summation range each (fixed trip counts, so the LDS reads pipeline;
    // thread-per-entry loops with data-dependent bounds cost 6 us each, wavefront-wide shuffles reductions 4 us)
    const int i = tid & (NB - 1), q = tid / NB;
    {
      double sum = 0.0;
#pragma unroll 8
      for (int kk = 0; kk < 32; ++kk) {
        const int k = 32 * q + kk;
        const double x = T[k * LDT + i];
        sum += (k <= i) ? x * cv[k] : 0.0;
      }
      T1[q * NB + i] = sum;
    }
    __syncthreads();
    if (tid < NB) {
      const double b = (T1[tid] + T1[NB + tid]) + (T1[2 * NB + tid] + T1[3 * NB + tid]);
      bv[tid] = b;
      f.bvec[tid] = b;
    }
    __syncthreads();
    {
      double sum = 0.0;
#pragma unroll 8
      for (int kk = 0; kk < 32; ++kk) {
        const int k = 32 * q + kk;
        const double x = T[i * LDT + k];
        sum += (k >= i) ? x * bv[k] : 0.0;
      }
      T1[q * NB + i] = sum;
    }
    __syncthreads();
    if (tid < NB) {
      const double t = (T1[tid] + T1[NB + tid]) + (T1[2 * NB + tid] + T1[3 * NB + tid]);
      tv[tid] = t;
      f.ttil[tid] = t;
    }
    __syncthreads();
    {
      double sum = 0.0;
#pragma unroll
      for (int kk = 0; kk < 32; ++kk) sum += ureg[kk] * tv[32 * q + kk];
      T1[q * NB + i] = sum;
    }
    __syncthreads();
    if (tid < NB) f.tvec[tid] = (T1[tid] + T1[NB + tid]) + (T1[2 * NB + tid] + T1[3 * NB + tid]);
    if (wid == 0) {
      double sum = log(rdiag[lane]) + log(rdiag[lane + 64]);
      for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
      if (lane == 0) f.logdet[0] = -2.0 * sum;
    } else if (wid == 1) {
      double sum = bv[lane] * bv[lane] + bv[lane + 64] * bv[lane + 64];
      for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
      if (lane == 0) f.bb[0] = sum;
    }
  }
}

__global__ __launch_bounds__(PT) void potrf_diag_kernel(double* __restrict__ A, int mp, int j,
                                                        double* __restrict__ dinv,
                                                        int* __restrict__ info, int flags, int m_real) {
  potrf_diag_body<0>(A, mp, j, dinv, info, flags, m_real, PotrfFuse{}, PotrfKm{});
}

__global__ __launch_bounds__(PT) void potrf_km_kernel(PotrfKm g, double* __restrict__ A, double* __restrict__ dinv,
                                                      int* __restrict__ info) {
  potrf_diag_body<2>(A, NB, 0, dinv, info, 64, g.m, PotrfFuse{}, g);
}

__global__ __launch_bounds__(PT) void potrf_fused_kernel(PotrfFuse f, double* __restrict__ A, double* __restrict__ dinv,
                                                         int* __restrict__ info, int m_real) {
  potrf_diag_body<1>(A, NB, 0, dinv, info, 64, m_real, f, PotrfKm{});
}

// ---- blocked factorisation without the engine: panel solve and trailing update of one 128-row step
//
// A step of the right-looking factorisation is a latency chain (diagonal block -> panel -> trailing blocks -> next
// diagonal block) of k = 128 products.  On the contraction engine such a launch costs 23-28 us whatever its size
// (workgroup entry, first-stage latency, 128 x 128 tiles on 4 wavefronts: 14 us of MFMAs alone), and the panel needed
// the full inverse of the diagonal block first (16 us + its store).  These two kernels are built for k = 128 instead:
// operands go from L2 straight into MFMA fragment registers, the tiles are small enough that every step spreads over
// the whole chip, and the panel is solved by substitution with the 16 x 16 micro inverses only.

// Panel: X = U_jj^-T A[j, c] in place for the column tiles c > j.  A workgroup takes 64 columns, a wavefront 16 of
// them -- columns are independent, so no barrier after the staging of U_jj.  With 16-row blocks b = 0..7:
//   X_b = D_b^T T_b,   T_b' -= U_bb'^T X_b  (b' > b),   D_b = inv(U_bb) from the factor kernel.
// An accumulator tile holds rows lq + 4r in register r: used as the B operand of the next MFMA it supplies the k index
// in that order, and the A operand (read from LDS) is indexed to match.
constexpr int PANEL_LDS = (36 + 8) * MB * MB * 8;
// With an identity right-hand side Y carried along (potrf_upper_blocked with Yinv), workgroups beyond the panel's own
// solve block row j of Y the same way: Y[j, c] <- U_jj^-T Y[j, c] for the column blocks c <= j that are non-zero, so that
// Y ends as U^-T and the separate triangular inversion goes away.
// Round 5: Y is never initialised and never transposed by launches of its own -- the block that starts as the identity
// (column block j of block row j) is not loaded, a block of Y is written back only while a later step reads it, and every
// solved block goes out a second time, transposed, as block (c, j) of X = U^-1 (X == null: Y is kept, X is not formed).
__global__ __launch_bounds__(256) void potrf_panel_kernel(double* __restrict__ A, int mp, int j,
                                                          const double* __restrict__ dmicro,
                                                          double* __restrict__ Y, double* __restrict__ X) {
  extern __shared__ __attribute__((aligned(16))) double L[];  // [36][16][16] blocks (i <= b) of U_jj, then [8][16][16] D_b
  double* Dm = L + 36 * MB * MB;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const double* Ujj = A + (int64_t)j * NB * mp + (int64_t)j * NB;
  {
    const int k = tid >> 4, m = tid & 15;
    int blk = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int b = i; b < 8; ++b, ++blk) L[blk * 256 + tid] = Ujj[(int64_t)(16 * i + k) * mp + 16 * b + m];
#pragma unroll
    for (int b = 0; b < 8; ++b) Dm[b * 256 + tid] = dmicro[b * 256 + tid];
  }
  const int na = (mp / NB - 1 - j) * 2;  // 64-column groups of the panel proper
  const bool rhs = (int)blockIdx.x >= na;
  const int c0 = rhs ? ((int)blockIdx.x - na) * 64 + wid * 16 : (j + 1) * NB + blockIdx.x * 64 + wid * 16;
  double* Ap = (rhs ? Y : A) + (int64_t)j * NB * mp + c0 + l15;
  pd4 T[8];
  const bool ident = rhs && X && c0 >= j * NB;  // (this block of Y is the identity so far)
#pragma unroll
  for (int b = 0; b < 8; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      T[b][r] = ident ? ((16 * b + lq + 4 * r == c0 - j * NB + l15) ? 1.0 : 0.0) : Ap[(int64_t)(16 * b + lq + 4 * r) * mp];
  __syncthreads();
  const int fo = lq * 16 + l15;  // fragment element of k-step r: [(lq + 4r)][l15] -> fo + 64 r
  const bool keep = !(rhs && X) || j + 1 < mp / NB;  // (the last block row of Y is read by nobody)
  int blk = 0;
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    pd4 x = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < 4; ++r) x = mfma_f64(Dm[b * 256 + fo + 64 * r], T[b][r], x);
    if (keep) {
#pragma unroll
      for (int r = 0; r < 4; ++r) Ap[(int64_t)(16 * b + lq + 4 * r) * mp] = x[r];
    }
    if (rhs && X) {
      double* Xp = X + (int64_t)(c0 + l15) * mp + (int64_t)j * NB + 16 * b + lq;
#pragma unroll
      for (int r = 0; r < 4; ++r) Xp[4 * r] = x[r];
    }
    ++blk;  // block (b, b) itself
#pragma unroll
    for (int b2 = b + 1; b2 < 8; ++b2, ++blk)
#pragma unroll
      for (int r = 0; r < 4; ++r) T[b2] = mfma_f64(-L[blk * 256 + fo + 64 * r], x[r], T[b2]);
  }
}

// Trailing update: A[r, c] -= X_r^T X_c over the upper 64 x 64 sub-tiles of the blocks behind step j (X = block row j,
// 128 deep).  One workgroup per sub-tile, a 32 x 32 quarter per wavefront; the fragments of both operands are rows of
// X, read from L2 as they are (16 consecutive doubles per k).
// ALL: every operand fragment is requested before the first MFMA (128 fragment registers, one wavefront per SIMD) -- the
// kernel's time is launch latency plus load round trips, and left alone the scheduler keeps only five k-steps of loads
// in flight (13 us of stalls for 3.4 us of MFMAs).  Used for steps of up to GPRHIP_POTRF_ALL_TILES sub-tiles (768:
// three rounds at that occupancy, measured to be as fast as or faster than the pipelined order up to there -- see
// potrf_upper_blocked); bigger steps run the compiler's pipelined order with four workgroups per CU.
// Workgroups beyond the symmetric part update the carried right-hand side: Y[r, c] -= X_r^T Y[j, c] for the row blocks
// r > j and the column blocks c <= j (nsym = number of symmetric sub-tiles; Y may be null).
template <bool ALL>
__device__ __forceinline__ void potrf_update_body(double* __restrict__ A, int mp, int j, double* __restrict__ Y,
                                                  int sym_lo, int sym_cnt, int y_lo, double* __restrict__ Xz) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const int ns = (mp / NB - 1 - j) * 2;
  const int64_t base = (int64_t)(j + 1) * NB;
  const double* X = A + (int64_t)j * NB * mp;
  int64_t r0, c0;
  const double* Xb;
  double* Cp;
  bool first = false;
  // the launch covers sub-tiles sym_lo .. sym_lo + sym_cnt - 1 of the symmetric list and the tiles of Y from y_lo on (a
  // whole step: 0, all, 0; the look-ahead split of potrf_upper_blocked: the two lists' prefixes = block row j + 1 first)
  if ((int)blockIdx.x < sym_cnt) {
    int si = 0, rem = sym_lo + (int)blockIdx.x;  // sub-tile (si <= sj) of the trailing part, in 64-blocks
    while (rem >= ns - si) {
      rem -= ns - si;
      ++si;
    }
    const int sj = si + rem;
    r0 = base + si * 64 + (wid >> 1) * 32;
    c0 = base + sj * 64 + (wid & 1) * 32;
    Xb = X + (int64_t)lq * mp + c0 + l15;
    Cp = A + (r0 + lq) * mp + c0 + l15;
  } else {
    const int t = y_lo + (int)blockIdx.x - sym_cnt, ncb = 2 * (j + 1);
    r0 = base + (t / ncb) * 64 + (wid >> 1) * 32;
    c0 = (int64_t)(t % ncb) * 64 + (wid & 1) * 32;
    Xb = Y + (int64_t)(j * NB + lq) * mp + c0 + l15;
    Cp = Y + (r0 + lq) * mp + c0 + l15;
    // first touch of this tile of Y (column block j, Xz given: Y is not initialised by a launch of its own): it starts as zero
    first = Xz != nullptr && c0 >= (int64_t)j * NB;
  }
  const double* Xa = X + (int64_t)lq * mp + r0 + l15;
  pd4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][jj][r] = first ? 0.0 : Cp[(int64_t)(16 * i + 4 * r) * mp + 16 * jj];
  if constexpr (ALL) {
    double av[NB / 4][2], bv[NB / 4][2];
#pragma unroll
    for (int ks = 0; ks < NB / 4; ++ks) {
      const int64_t o = (int64_t)(4 * ks) * mp;
      av[ks][0] = Xa[o];
      av[ks][1] = Xa[o + 16];
      bv[ks][0] = Xb[o];
      bv[ks][1] = Xb[o + 16];
    }
#pragma unroll
    for (int ks = 0; ks < NB / 4; ++ks) {
      const double a0 = -av[ks][0], a1 = -av[ks][1];
      acc[0][0] = mfma_f64(a0, bv[ks][0], acc[0][0]);
      acc[0][1] = mfma_f64(a0, bv[ks][1], acc[0][1]);
      acc[1][0] = mfma_f64(a1, bv[ks][0], acc[1][0]);
      acc[1][1] = mfma_f64(a1, bv[ks][1], acc[1][1]);
    }
  } else {
#pragma unroll 8
    for (int ks = 0; ks < NB / 4; ++ks) {
      const int64_t o = (int64_t)(4 * ks) * mp;
      const double a0 = -Xa[o], a1 = -Xa[o + 16], b0 = Xb[o], b1 = Xb[o + 16];
      acc[0][0] = mfma_f64(a0, b0, acc[0][0]);
      acc[0][1] = mfma_f64(a0, b1, acc[0][1]);
      acc[1][0] = mfma_f64(a1, b0, acc[1][0]);
      acc[1][1] = mfma_f64(a1, b1, acc[1][1]);
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int r = 0; r < 4; ++r) Cp[(int64_t)(16 * i + 4 * r) * mp + 16 * jj] = acc[i][jj][r];
  if (first) {  // ... and the block of U^-1 below the diagonal that mirrors it is zero
    double* Zp = Xz + (r0 + lq) * mp + c0 + l15;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int r = 0; r < 4; ++r) Zp[(int64_t)(16 * i + 4 * r) * mp + 16 * jj] = 0.0;
  }
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void potrf_update_all_kernel(
    double* __restrict__ A, int mp, int j, double* __restrict__ Y, int sym_lo, int sym_cnt, int y_lo, double* __restrict__ Xz) {
  potrf_update_body<true>(A, mp, j, Y, sym_lo, sym_cnt, y_lo, Xz);
}
__global__ __launch_bounds__(256) void potrf_update_kernel(double* __restrict__ A, int mp, int j, double* __restrict__ Y,
                                                           int sym_lo, int sym_cnt, int y_lo, double* __restrict__ Xz) {
  potrf_update_body<false>(A, mp, j, Y, sym_lo, sym_cnt, y_lo, Xz);
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

__global__ void scatter_diag_blocks_kernel(const double* __restrict__ dinv, int mp, double* __restrict__ X) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int r = blockIdx.y;
  if (c >= mp) return;
  const int br = r / NB, bc = c / NB;
  X[(int64_t)r * mp + c] = (br == bc) ? dinv[(int64_t)br * NB * NB + (r % NB) * NB + (c % NB)] : 0.0;
}

void launch_scatter_diag_blocks(const double* dinv, int mp, double* X, hipStream_t s) {
  hipLaunchKernelGGL(scatter_diag_blocks_kernel, dim3((mp + 255) / 256, mp), dim3(256), 0, s, dinv, mp, X);
  GPR_HIP(hipGetLastError());
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

#ifdef GPRHIP_LAB  // (measured 1.4 - 2.8 x slower than the step launches, DESIGN section 4: lab build only)
// ---- the whole factorisation + carried inverse as ONE launch with device-side dependencies (round 5) ---------------
//
// potrf_upper_blocked above is a chain of three launches per 128-row step (diagonal block -> panel -> trailing update),
// ~58 us per step at any m: each launch pays its own entry, its operands' first round trip and its drain, and nothing of
// step j + 1 can start before the last tile of step j has been written.  Here the same tile tasks run inside one
// persistent kernel and wait for exactly what they read:
//   * the first workgroup to arrive walks the diagonal blocks: D(j) = factor of block (j, j) in LDS (potrf_factor_lds);
//   * every other workgroup takes tile tasks off ONE ticket counter (an atomic add: a first version handed out two lists
//     through compare-and-swap with an eligibility test, and the 255 contending workgroups serialised on it at 1.35 us
//     per task -- 8 ms at m = 2048).  The list is ordered so that the diagonal chain runs ahead of the bulk of the
//     trailing updates (look-ahead) instead of waiting for all of them:
//         crit(0) | for j = 0, 1, ..:  U(j, block row j + 2) | crit(j + 1) | U(j, block rows j + 3 ..), PY(j, .), UY(j, ., .)
//     with crit(j) = the panel blocks P(j, c) and the updates of block row j + 1 (what the next diagonal block and the next
//     panels wait for); PY: Y[j, c] <- U_jj^-T Y[j, c] of the carried right-hand side, which also stores its transpose as
//     block (c, j) of U^-1; UY: Y[r, c] -= U[j, r]^T Y[j, c];
//   * a task waits (one thread polls, bounded) for the flags of the tasks it reads from: diag[j], panel[j][c] and a
//     version count per 64 x 64 sub-tile (= the number of steps applied to it), all tagged with the launch's epoch so
//     nothing has to be cleared between launches.
// The order is topological (every task comes behind the tasks it waits for, the diagonal blocks aside, whose workgroup
// takes no tickets), and tickets go out in order: whatever a running task waits for is running or done -- no deadlock
// whatever the number of resident workgroups.  Tile data that crosses workgroups (A, Y, the micro
// inverses) moves between a release fence (before a flag is set) and an acquire fence (after it has been seen) at agent
// scope: the eight XCDs' L2s are not coherent with each other.
// A poll that exceeds its bound sets the abort word and every workgroup leaves: *info = POTRF_CHAIN_ABORT.
namespace {

enum { CT_P = 0, CT_PY = 1, CT_U = 2, CT_UY = 3 };
struct ChainTask {
  int kind, j, a, b;  // P/PY: a = column block;  U/UY: a = first 64-row sub-tile, b = 64-column sub-tile | count << 16
};
enum { CS_TICKET = 0, CS_ARRIVE = 2, CS_DONE = 3, CS_ABORT = 4, CS_HDR = 8 };
constexpr int CHAIN_SPIN_LIMIT = 1 << 22;

// Tile data that crosses workgroups (A, Y, the micro inverses) moves with plain loads and stores between an acquire and a
// release fence at agent scope (the eight XCDs' L2s are not coherent with each other: the release writes the L2's dirty
// lines back, the acquire drops its stale ones).  A first version used agent-scope atomic loads / stores for every element
// instead (no fences): bit-identical results, but such accesses are not coalesced -- 8 ms instead of 0.9 at m = 2048.
__device__ __forceinline__ double ld_coh(const double* p) { return *p; }
__device__ __forceinline__ void st_coh(double* p, double v) { *p = v; }
__device__ __forceinline__ void chain_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
__device__ __forceinline__ void chain_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); }
__device__ __forceinline__ int ld_flag(const int* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_flag(int* p, int v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct ChainArgs {
  double* A;
  double* Y;
  double* X;      // U^-1 (output)
  double* dinv;   // [nb][128][128] slots; the head of slot j receives the eight micro inverses of block j
  int* info;
  int* sync;      // header | diag[nb] | panel[nb*nb] | panely[nb*nb] | ver[(2nb)^2] | very[(2nb)^2]
  const ChainTask* tasks;
  int ntasks;
  unsigned long long* trace;  // tools/potrf_check: [ntasks + nb][4] = {start, dependencies seen, work done, flag set} in 10 ns ticks, or null
  int mp, nb, m_real, epoch;
};

// one thread: wait until *p == want (tagged); false on abort
__device__ __forceinline__ bool chain_wait(const int* p, int want, int* sync, int epoch) {
  for (int it = 0; it < CHAIN_SPIN_LIMIT; ++it) {
    if (ld_flag(p) == want) return true;
    if ((it & 63) == 63 && ld_flag(sync + CS_ABORT) == epoch) return false;
    __builtin_amdgcn_s_sleep(1);
  }
  st_flag(sync + CS_ABORT, epoch);
  return false;
}

}  // namespace

constexpr int POTRF_CHAIN_ABORT = POTRF_CHAIN_ABORT_CODE;

__global__ __launch_bounds__(PT) void potrf_chain_kernel(ChainArgs g) {
  extern __shared__ __attribute__((aligned(16))) double T[];  // diag: [NB][LDT] | T1 | rdiag | ...;  panel: [36+8][256]
  __shared__ int s_role, s_task, s_ok;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const int nb = g.nb, mp = g.mp, n2 = 2 * nb;
  int* const diag = g.sync + CS_HDR;
  int* const panel = diag + nb;
  int* const panely = panel + nb * nb;
  int* const ver = panely + nb * nb;
  int* const very = ver + n2 * n2;
  const int tag = g.epoch << 8;  // versions are small counts (<= nb <= 64)
  if (tid == 0) s_role = atomicAdd(g.sync + CS_ARRIVE, 1);
  __syncthreads();
  const bool diag_role = s_role == 0;

  if (diag_role) {
    // ---------------- the diagonal chain
    double* T1 = T + NB * LDT;
    double* rdiag = T1 + NB * MB;
    int& bad = *reinterpret_cast<int*>(rdiag + 4 * NB);
    for (int j = 0; j < nb; ++j) {
      unsigned long long* const tr = g.trace ? g.trace + 4 * (int64_t)(g.ntasks + j) : nullptr;
      if (tid == 0) {
        if (tr) tr[0] = wall_clock64();
        bool ok = true;
        if (j > 0) {
          ok = chain_wait(ver + (2 * j) * n2 + 2 * j, tag | j, g.sync, g.epoch) &&
               chain_wait(ver + (2 * j) * n2 + 2 * j + 1, tag | j, g.sync, g.epoch) &&
               chain_wait(ver + (2 * j + 1) * n2 + 2 * j + 1, tag | j, g.sync, g.epoch);
        }
        s_ok = ok;
        bad = 0;
        if (tr) tr[1] = wall_clock64();
      }
      __syncthreads();
      if (!s_ok) break;
      chain_acquire();
      const int live = (g.m_real > 0) ? min(NB, max(1, g.m_real - j * NB)) : NB;
      const int k_end = (live + MB - 1) / MB * MB;
      double* Ab = g.A + (int64_t)j * NB * mp + (int64_t)j * NB;
      for (int idx = tid; idx < NB * NB; idx += PT) {
        const int r = idx >> 7, c = idx & (NB - 1);
        if (c >= r) T[r * LDT + c] = ld_coh(Ab + (int64_t)r * mp + c);
      }
      if (tid < NB && tid >= k_end) rdiag[tid] = 1.0;
      __syncthreads();
      potrf_factor_lds(T, rdiag, &bad, k_end, false, false);
      if (tid == 0 && bad != 0) atomicCAS(g.info, 0, j * NB + bad);
      // U_jj back (strict lower of the block zeroed) and the eight micro inverses [8][16][16] at the head of slot j
      for (int idx = tid; idx < NB * NB; idx += PT) {
        const int r = idx >> 7, c = idx & (NB - 1);
        st_coh(Ab + (int64_t)r * mp + c, c >= r ? T[r * LDT + c] : 0.0);
      }
      double* dj = g.dinv + (int64_t)j * NB * NB;
      for (int idx = tid; idx < 8 * MB * MB; idx += PT)
        st_coh(dj + idx, dblk(T, idx >> 8)[((idx >> 4) & 15) * LDT + (idx & 15)]);
      if (tr && tid == 0) tr[2] = wall_clock64();
      chain_release();
      __syncthreads();
      if (tid == 0) {
        st_flag(diag + j, g.epoch);
        if (tr) tr[3] = wall_clock64();
      }
    }
  } else {
    // ---------------- tile tasks
    for (;;) {
      if (tid == 0) {
        int pick = -1;  // -1: nothing left (or aborted)
        if (ld_flag(g.sync + CS_ABORT) != g.epoch) {
          const int tk = atomicAdd(g.sync + CS_TICKET, 1);
          if (tk < g.ntasks) pick = tk;
        }
        s_task = pick;
      }
      __syncthreads();
      const int pick = s_task;
      if (pick == -1) break;
      const ChainTask t = g.tasks[pick];
      const int j = t.j;
      unsigned long long* const tr = g.trace ? g.trace + 4 * (int64_t)pick : nullptr;
      if (tr && tid == 0) tr[0] = wall_clock64();
      if (t.kind == CT_P || t.kind == CT_PY) {
        const bool rhs = t.kind == CT_PY;
        const int c = t.a;
        double* const M = rhs ? g.Y : g.A;
        if (tid == 0) {
          bool ok = chain_wait(diag + j, g.epoch, g.sync, g.epoch);
          // the block's four sub-tiles carry every earlier step: j of them (A), j - c of them (Y; none for the block
          // that starts as the identity)
          const int want = rhs ? j - c : j;
          int* const vv = rhs ? very : ver;
          if (want > 0)
            for (int q = 0; q < 4 && ok; ++q) {
              const int r64 = 2 * j + (q >> 1), c64 = 2 * c + (q & 1);
              if (!rhs && r64 > c64) continue;  // (cannot happen: c > j)
              ok = chain_wait(vv + r64 * n2 + c64, tag | want, g.sync, g.epoch);
            }
          s_ok = ok;
          if (tr) tr[1] = wall_clock64();
        }
        __syncthreads();
        if (!s_ok) break;
        chain_acquire();
        // stage the blocks (i <= b) of U_jj and the micro inverses; a wavefront owns 16 columns of the block row
        double* L = T;
        double* Dm = T + 36 * MB * MB;
        const double* Ujj = g.A + (int64_t)j * NB * mp + (int64_t)j * NB;
        const double* dmicro = g.dinv + (int64_t)j * NB * NB;
        for (int e = tid; e < 36 * 256; e += PT) {
          const int blk = e >> 8, w = e & 255, k = w >> 4, mm = w & 15;
          int i = 0, rem = blk;  // blk -> (i, b), i <= b, row-major over the upper triangle of the 8 x 8 block grid
          while (rem >= 8 - i) {
            rem -= 8 - i;
            ++i;
          }
          const int b = i + rem;
          L[e] = ld_coh(Ujj + (int64_t)(16 * i + k) * mp + 16 * b + mm);
        }
        for (int e = tid; e < 8 * 256; e += PT) Dm[e] = ld_coh(dmicro + e);
        const int c0 = c * NB + wid * 16;
        double* Ap = M + (int64_t)j * NB * mp + c0 + l15;
        pd4 Tt[8];
        const bool ident = rhs && c == j;
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            Tt[b][r] = ident ? ((16 * b + lq + 4 * r == wid * 16 + l15) ? 1.0 : 0.0) : ld_coh(Ap + (int64_t)(16 * b + lq + 4 * r) * mp);
        __syncthreads();
        const int fo = lq * 16 + l15;
        const bool keep = !rhs || j + 1 < nb;  // (the last block row of Y is read by nobody)
        int blk = 0;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          pd4 x = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int r = 0; r < 4; ++r) x = mfma_f64(Dm[b * 256 + fo + 64 * r], Tt[b][r], x);
          if (keep) {
#pragma unroll
            for (int r = 0; r < 4; ++r) st_coh(Ap + (int64_t)(16 * b + lq + 4 * r) * mp, x[r]);
          }
          if (rhs) {  // block (c, j) of U^-1 = this block transposed
            double* Xp = g.X + (int64_t)(c0 + l15) * mp + (int64_t)j * NB + 16 * b + lq;
#pragma unroll
            for (int r = 0; r < 4; ++r) Xp[4 * r] = x[r];
          }
          ++blk;
#pragma unroll
          for (int b2 = b + 1; b2 < 8; ++b2, ++blk)
#pragma unroll
            for (int r = 0; r < 4; ++r) Tt[b2] = mfma_f64(-L[blk * 256 + fo + 64 * r], x[r], Tt[b2]);
        }
        if (tr && tid == 0) tr[2] = wall_clock64();
        chain_release();
        __syncthreads();
        if (tid == 0) {
          st_flag((rhs ? panely : panel) + j * nb + c, g.epoch);
          if (tr) tr[3] = wall_clock64();
        }
      } else {
        // trailing update: one 64 x 64 sub-tile per group of four wavefronts, a 32 x 32 quarter per wavefront
        const bool rhs = t.kind == CT_UY;
        const int c64 = t.b & 0xffff, cnt = t.b >> 16;
        const int grp = wid >> 2, w4 = wid & 3;
        const int cblk = c64 >> 1;
        if (tid == 0) {
          bool ok = true;
          for (int q = 0; q < cnt && ok; ++q) ok = chain_wait(panel + j * nb + ((t.a + q) >> 1), g.epoch, g.sync, g.epoch);
          if (ok) ok = chain_wait((rhs ? panely : panel) + j * nb + cblk, g.epoch, g.sync, g.epoch);
          const int want = rhs ? j - cblk : j;
          int* const vv = rhs ? very : ver;
          if (want > 0)
            for (int q = 0; q < cnt && ok; ++q) ok = chain_wait(vv + (t.a + q) * n2 + c64, tag | want, g.sync, g.epoch);
          s_ok = ok;
          if (tr) tr[1] = wall_clock64();
        }
        __syncthreads();
        if (!s_ok) break;
        chain_acquire();
        const int r64 = t.a + grp;
        if (grp < cnt) {
          const double* Xr = g.A + (int64_t)j * NB * mp;  // block row j of the factor
          const int64_t r0 = (int64_t)r64 * 64 + (w4 >> 1) * 32, c0 = (int64_t)c64 * 64 + (w4 & 1) * 32;
          const double* Xa = Xr + (int64_t)lq * mp + r0 + l15;
          const double* Xb = (rhs ? g.Y + (int64_t)j * NB * mp : Xr) + (int64_t)lq * mp + c0 + l15;
          double* Cp = (rhs ? g.Y : g.A) + (r0 + lq) * mp + c0 + l15;
          const bool first = rhs && j == cblk;  // first touch of this sub-tile of Y: it starts as zero
          pd4 acc[2][2];
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
              for (int r = 0; r < 4; ++r) acc[i][jj][r] = first ? 0.0 : ld_coh(Cp + (int64_t)(16 * i + 4 * r) * mp + 16 * jj);
          double av[NB / 4][2], bv[NB / 4][2];
#pragma unroll
          for (int ks = 0; ks < NB / 4; ++ks) {
            const int64_t o = (int64_t)(4 * ks) * mp;
            av[ks][0] = ld_coh(Xa + o);
            av[ks][1] = ld_coh(Xa + o + 16);
            bv[ks][0] = ld_coh(Xb + o);
            bv[ks][1] = ld_coh(Xb + o + 16);
          }
#pragma unroll
          for (int ks = 0; ks < NB / 4; ++ks) {
            const double a0 = -av[ks][0], a1 = -av[ks][1];
            acc[0][0] = mfma_f64(a0, bv[ks][0], acc[0][0]);
            acc[0][1] = mfma_f64(a0, bv[ks][1], acc[0][1]);
            acc[1][0] = mfma_f64(a1, bv[ks][0], acc[1][0]);
            acc[1][1] = mfma_f64(a1, bv[ks][1], acc[1][1]);
          }
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
              for (int r = 0; r < 4; ++r) st_coh(Cp + (int64_t)(16 * i + 4 * r) * mp + 16 * jj, acc[i][jj][r]);
          if (first) {  // ... and the block of U^-1 below the diagonal that mirrors it is zero
            double* Zp = g.X + (r0 + lq) * mp + c0 + l15;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int r = 0; r < 4; ++r) Zp[(int64_t)(16 * i + 4 * r) * mp + 16 * jj] = 0.0;
          }
        }
        if (tr && tid == 0) tr[2] = wall_clock64();
        chain_release();
        __syncthreads();
        if (tid == 0) {
          int* const vv = rhs ? very : ver;
          const int now = (rhs ? j - cblk : j) + 1;
          for (int q = 0; q < cnt; ++q) st_flag(vv + (t.a + q) * n2 + c64, tag | now);
          if (tr) tr[3] = wall_clock64();
        }
      }
    }
  }
  // the last workgroup out resets the tickets for the next launch; an aborted launch reports itself
  __syncthreads();
  if (tid == 0) {
    if (ld_flag(g.sync + CS_ABORT) == g.epoch) atomicExch(g.info, POTRF_CHAIN_ABORT);
    __threadfence();
    if (atomicAdd(g.sync + CS_DONE, 1) == (int)gridDim.x - 1) {
      st_flag(g.sync + CS_TICKET, 0);
      st_flag(g.sync + CS_ARRIVE, 0);
      st_flag(g.sync + CS_DONE, 0);
    }
  }
}

#endif  // GPRHIP_LAB

// dynamic-LDS opt-in of the two step kernels, once per device
static void potrf_attrs() {
  static uint64_t done = 0;
  once_per_device(done, [] {
    GPR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&potrf_diag_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, POTRF_LDS));
    GPR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&potrf_fused_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, POTRF_LDS));
    GPR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&potrf_km_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, POTRF_LDS));
    GPR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&potrf_panel_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, PANEL_LDS));
#ifdef GPRHIP_LAB
    GPR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&potrf_chain_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, POTRF_LDS));
#endif
  });
}
void launch_potrf_km(const PotrfKm& g, double* A, double* Xinv, int* info, hipStream_t s) {
  potrf_attrs();
  hipLaunchKernelGGL(potrf_km_kernel, dim3(1), dim3(PT), POTRF_LDS, s, g, A, Xinv, info);
  GPR_HIP(hipGetLastError());
}
void launch_potrf_fused(const PotrfFuse& f, double* A, double* Xinv, int* info, int m_real, hipStream_t s) {
  potrf_attrs();
  hipLaunchKernelGGL(potrf_fused_kernel, dim3(1), dim3(PT), POTRF_LDS, s, f, A, Xinv, info, m_real);
  GPR_HIP(hipGetLastError());
}
void potrf_fetch_timestamps(unsigned long long* out64) {
  GPR_HIP(hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_potrf_ts), 64 * sizeof(unsigned long long)));
}
void launch_potrf_diag(double* A, int mp, int j, double* dinv, int* info, hipStream_t s) {
  launch_potrf_diag_flags(A, mp, j, dinv, info, 0, s, 0);
}
// flags: ablation switches of tools/potrf_check (bit0 skip factor, bit1 skip invert); 0 in the library
void launch_potrf_diag_flags(double* A, int mp, int j, double* dinv, int* info, int flags, hipStream_t s, int m_real) {
  potrf_attrs();
  hipLaunchKernelGGL(potrf_diag_kernel, dim3(1), dim3(PT), POTRF_LDS, s, A, mp, j, dinv, info, flags, m_real);
  GPR_HIP(hipGetLastError());
}

// Blocked upper Cholesky A = U^T U in place (dpotrf `U; lib/fitc_gp.ml:56) with inv(U_jj) of every diagonal block in
// dinv: per step a factor-only diagonal kernel, the substitution panel and the small-tile trailing update; the block
// inverses (which nothing on the chain needs any more) are formed by one launch over all blocks at the end.
void potrf_upper_blocked(hipStream_t s, double* A, int mp, double* dinv, int* info, double* Yscratch, double* Xinv,
                         int m_real, const PotrfAux* aux) {
  potrf_attrs();
  static const int all_tiles = [] {  // largest step (in 64 x 64 sub-tiles) that runs the all-loads-first update kernel
    const char* e = getenv("GPRHIP_POTRF_ALL_TILES");
    return e ? atoi(e) : 768;
  }();
  // aux->min_rest: smallest rest of a step (sub-tiles outside block row j + 1) that goes to the side stream; 0 / no aux =
  // never, the library's default: measured SLOWER than one stream -- m = 2048: 946 against 856 us, m = 1024: 441 against
  // 384 us, tools/potrf_check, profiles/r05_potrf_lookahead.txt -- the two cross-stream event waits of a step cost more
  // than the part of the update they take off the chain
  const int la_min = aux ? aux->min_rest : 0;
  const int nb = mp / NB;
  if (nb == 1) {  // a single block: factor and inverse in one launch
    // (a single block's inverse, [128][128], IS the mp x mp inverse: written straight to Xinv)
    hipLaunchKernelGGL(potrf_diag_kernel, dim3(1), dim3(PT), POTRF_LDS, s, A, mp, 0, Xinv ? Xinv : dinv, info, 64, m_real);
    GPR_HIP(hipGetLastError());
    return;
  }
  // With Xinv the inverse of the factor comes out of the same steps: an identity right-hand side Y rides along (panel:
  // Y[j, :] <- U_jj^-T Y[j, :], update: Y[r, :] -= U[j, r]^T Y[j, :]), Y ends as U^-T and X = Y^T.  The extra tiles run
  // on compute units the latency-bound steps leave idle; the recursive-doubling inversion (eight engine launches,
  // 0.29 ms at m = 2048) and the block-inverse launch are not needed then.
  if (Xinv && !Yscratch) {
    set_error("gprhip: potrf_upper_blocked: the inverse needs an mp x mp scratch");
    throw HipFail{ST_BAD_ARG};
  }
  double* const Y = Xinv ? Yscratch : nullptr;
  // (round 5: no launch sets Y to the identity and none transposes it at the end -- 5 us each on the chain of every
  // factorisation: the panel kernel starts the diagonal blocks of Y from the identity and writes every solved block
  // straight into X = U^-1, transposed; the update kernel's first touch of a tile of Y starts from zero and clears the
  // mirrored tile of X)
  auto update = [&](hipStream_t st, int j, int sym_lo, int sym_cnt, int y_lo, int y_cnt, int step_tiles) {
    if (sym_cnt + y_cnt <= 0) return;
    if (step_tiles <= all_tiles)
      hipLaunchKernelGGL(potrf_update_all_kernel, dim3(sym_cnt + y_cnt), dim3(256), 0, st, A, mp, j, Y, sym_lo, sym_cnt, y_lo, Xinv);
    else
      hipLaunchKernelGGL(potrf_update_kernel, dim3(sym_cnt + y_cnt), dim3(256), 0, st, A, mp, j, Y, sym_lo, sym_cnt, y_lo, Xinv);
  };
  // Look-ahead (round 5; built, measured, off by default -- see la_min above).  The next diagonal block, the next panel and the next step's share of Y read block row j + 1
  // of the trailing update only -- the first 2 ns - 1 sub-tiles of its symmetric list and the first two sub-tile rows of
  // its Y part.  With a side stream those go out alone on `s` (a launch of a few dozen workgroups) and the rest of the
  // step's update runs on the side stream beside the next diagonal block's factorisation (30 us of one CU); `s` waits for
  // it only in front of the NEXT step's update, which touches the same tiles.  Per step on `s`: diag + panel + the short
  // update instead of diag + panel + the whole update (11 - 17 us at m = 2048, 20 - 40 us at m = 4096).
  bool pending = false;  // a rest launch of the previous step is in flight on the side stream
  for (int j = 0; j < nb; ++j) {
    double* dj = dinv + (int64_t)j * NB * NB;
    hipLaunchKernelGGL(potrf_diag_kernel, dim3(1), dim3(PT), POTRF_LDS, s, A, mp, j, dj, info, 2 + 32, m_real);
    const int rest = nb - 1 - j, ns = 2 * rest;
    const int nrhs = Y ? 2 * (j + 1) : 0;  // 64-column groups of the right-hand side that are non-zero in block row j
    if (rest + nrhs > 0)
      hipLaunchKernelGGL(potrf_panel_kernel, dim3(2 * rest + nrhs), dim3(256), PANEL_LDS, s, A, mp, j, dj, Y, Xinv);
    if (rest > 0) {
      const int nsym = ns * (ns + 1) / 2, ny = ns * nrhs, tiles = nsym + ny;
      const int csym = std::min(nsym, 2 * ns - 1), cy = std::min(ny, 2 * nrhs);  // block row j + 1
#ifdef GPRHIP_LAB  // (two-stream look-ahead: measured slower wherever it engages, lab build only)
      const bool split = aux && aux->side && la_min > 0 && (nsym - csym) + (ny - cy) >= la_min;
#else
      const bool split = false;
#endif
      if (pending) {  // this step's update writes the tiles the previous step's rest is still writing
        GPR_HIP(hipStreamWaitEvent(s, aux->ev_rest[(j - 1) & 1], 0));
        pending = false;
      }
      if (split) {
        GPR_HIP(hipEventRecord(aux->ev_panel[j & 1], s));
        GPR_HIP(hipStreamWaitEvent(aux->side, aux->ev_panel[j & 1], 0));
        update(aux->side, j, csym, nsym - csym, cy, ny - cy, tiles);
        GPR_HIP(hipEventRecord(aux->ev_rest[j & 1], aux->side));
        pending = true;
        update(s, j, 0, csym, 0, cy, tiles);
      } else {
        update(s, j, 0, nsym, 0, ny, tiles);
      }
    }
  }
  if (pending) GPR_HIP(hipStreamWaitEvent(s, aux->ev_rest[(nb - 2) & 1], 0));
  if (!Y) hipLaunchKernelGGL(potrf_diag_kernel, dim3(nb), dim3(PT), POTRF_LDS, s, A, mp, 0, dinv, info, 1, m_real);
  GPR_HIP(hipGetLastError());
}

#ifdef GPRHIP_LAB
struct PotrfChain {
  int device = 0, mp = 0, nb = 0, ntasks = 0, epoch = 0, grid = 0;
  int* sync = nullptr;
  ChainTask* tasks = nullptr;
};

void potrf_chain_destroy(PotrfChain* ch);
// Task list of an nb-block factorisation with the carried inverse (host side; the order is explained at the kernel).
PotrfChain* potrf_chain_create(int mp) {
  const int nb = mp / NB, n2 = 2 * nb;
  if (mp % NB || nb < 2 || nb > 64) return nullptr;
  std::vector<ChainTask> tasks;
  auto crit = [&](int j) {  // panels of step j, then the updates of block row j + 1 (one sub-tile per task, the next
    if (j >= nb) return;    // diagonal block's three first)
    for (int c = j + 1; c < nb; ++c) tasks.push_back({CT_P, j, c, 0});
    if (j + 1 >= nb) return;
    for (int r64 = 2 * j + 2; r64 <= 2 * j + 3; ++r64)
      for (int c64 = r64; c64 <= 2 * j + 3; ++c64) tasks.push_back({CT_U, j, r64, c64 | (1 << 16)});
    for (int c64 = 2 * j + 4; c64 < n2; ++c64)
      for (int r64 = 2 * j + 2; r64 <= 2 * j + 3; ++r64) tasks.push_back({CT_U, j, r64, c64 | (1 << 16)});
  };
  auto u_rows = [&](int j, int R) {  // trailing update of step j on the 128-row block R (two sub-tiles per task)
    for (int c64 = 2 * R; c64 < n2; ++c64) tasks.push_back({CT_U, j, 2 * R, c64 | ((c64 == 2 * R ? 1 : 2) << 16)});
  };
  crit(0);
  for (int j = 0; j < nb; ++j) {
    if (j + 2 < nb) u_rows(j, j + 2);  // what the crit tasks of step j + 1 read
    crit(j + 1);
    for (int R = j + 3; R < nb; ++R) u_rows(j, R);
    for (int c = j; c >= 0; --c) tasks.push_back({CT_PY, j, c, 0});
    for (int R = j + 1; R < nb; ++R)
      for (int c64 = 0; c64 <= 2 * j + 1; ++c64) tasks.push_back({CT_UY, j, 2 * R, c64 | (2 << 16)});
  }
  struct Free {  // (a failed allocation half-way must not leak the buffers before it)
    void operator()(PotrfChain* c) const { potrf_chain_destroy(c); }
  };
  std::unique_ptr<PotrfChain, Free> ch(new PotrfChain());
  GPR_HIP(hipGetDevice(&ch->device));
  ch->mp = mp;
  ch->nb = nb;
  ch->ntasks = (int)tasks.size();
  int cus = 0;
  GPR_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ch->device));
  // Workgroups: the diagonal chain + enough tile workers to keep the trailing updates of a step inside the time the chain
  // needs for it (8 per 128-block measured sufficient up to m = 2048), not the whole chip: every workgroup pins 150 KB of
  // LDS, and the covariance of the first row chunk runs beside the K_m factorisation on the problem's second stream.
  int want = 8 * nb;
  if (const char* e = getenv("GPRHIP_CHAIN_GRID")) want = atoi(e);
  ch->grid = std::max(2, std::min(std::min(cus > 0 ? cus : 256, std::max(32, want)), 1 + ch->ntasks));
  const size_t nsync = CS_HDR + nb + 2 * (size_t)nb * nb + 2 * (size_t)n2 * n2;
  GPR_HIP(hipMalloc(&ch->sync, nsync * sizeof(int)));
  GPR_HIP(hipMemset(ch->sync, 0, nsync * sizeof(int)));
  GPR_HIP(hipMalloc(&ch->tasks, tasks.size() * sizeof(ChainTask)));
  GPR_HIP(hipMemcpy(ch->tasks, tasks.data(), tasks.size() * sizeof(ChainTask), hipMemcpyHostToDevice));
  return ch.release();
}

void potrf_chain_destroy(PotrfChain* ch) {
  if (!ch) return;
  (void)hipFree(ch->sync);
  (void)hipFree(ch->tasks);
  delete ch;
}

// A = U^T U in place (upper) and Xinv = U^-1 in one launch; Y: mp x mp scratch (the carried right-hand side); dinv: scratch
// for the micro inverses ([mp/128][128][128] as for potrf_upper_blocked).  *info: first non-positive pivot (1-based) or
// POTRF_CHAIN_ABORT.
int potrf_chain_tasks(const PotrfChain* ch, int* kinds4) {
  if (ch && kinds4) GPR_HIP(hipMemcpy(kinds4, ch->tasks, (size_t)ch->ntasks * sizeof(ChainTask), hipMemcpyDeviceToHost));
  return ch ? ch->ntasks : 0;
}

void potrf_upper_chain(hipStream_t s, PotrfChain* ch, double* A, int mp, double* dinv, int* info, double* Y, double* Xinv,
                       int m_real, unsigned long long* trace) {
  potrf_attrs();
  if (!ch || ch->mp != mp || !Y || !Xinv) {
    set_error("gprhip: potrf_upper_chain: workspace does not match the matrix");
    throw HipFail{ST_BAD_ARG};
  }
  if (++ch->epoch >= (1 << 22)) {  // tags are epoch << 8: start over long before they wrap
    GPR_HIP(hipStreamSynchronize(s));
    const size_t nsync = CS_HDR + ch->nb + 2 * (size_t)ch->nb * ch->nb + 8 * (size_t)ch->nb * ch->nb;
    GPR_HIP(hipMemset(ch->sync, 0, nsync * sizeof(int)));
    ch->epoch = 1;
  }
  ChainArgs g;
  g.A = A; g.Y = Y; g.X = Xinv; g.dinv = dinv; g.info = info; g.sync = ch->sync;
  g.tasks = ch->tasks; g.ntasks = ch->ntasks; g.trace = trace;
  g.mp = mp; g.nb = ch->nb; g.m_real = m_real; g.epoch = ch->epoch;
  hipLaunchKernelGGL(potrf_chain_kernel, dim3(ch->grid), dim3(PT), POTRF_LDS, s, g);
  GPR_HIP(hipGetLastError());
}

#endif  // GPRHIP_LAB

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

void launch_triu_matvec(const double* A, int mp, const double* x, double* y, int trans, hipStream_t s) {
  hipLaunchKernelGGL(triu_matvec_kernel, dim3((mp + 3) / 4), dim3(256), 0, s, A, mp, x, y, trans);
  GPR_HIP(hipGetLastError());
}

// The same product with a rider: one more workgroup of the launch forms a scalar that would otherwise be a 5 us launch of
// its own on the latency chain between the B~ factorisation and the Q' products --
//   rider 1: rout[0] = 2 sum_{i < rn} log rA[i (mp + 1)]   (log determinant from a factor's diagonal, lib/utils.ml:95-101)
//   rider 2: rout[0] = sum_{i < rn} rA[i]^2                 (|b|^2 of a vector an EARLIER launch finished)
__global__ __launch_bounds__(256) void triu_matvec_rider_kernel(const double* __restrict__ A, int mp,
                                                                const double* __restrict__ x, double* __restrict__ y,
                                                                int trans, int rider, const double* __restrict__ rA, int rn,
                                                                double* __restrict__ rout) {
  if (blockIdx.x == gridDim.x - 1) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < rn; i += 256) s += rider == 1 ? log(rA[(int64_t)i * (mp + 1)]) : rA[i] * rA[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) rout[0] = rider == 1 ? red[0] + red[0] : red[0];
    return;
  }
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
void launch_triu_matvec_rider(const double* A, int mp, const double* x, double* y, int trans, int rider, const double* rA,
                              int rn, double* rout, hipStream_t s) {
  hipLaunchKernelGGL(triu_matvec_rider_kernel, dim3((mp + 3) / 4 + 1), dim3(256), 0, s, A, mp, x, y, trans, rider, rA, rn, rout);
  GPR_HIP(hipGetLastError());
}

}  // namespace gprhip
