// MFMA contraction engine (see mfma_gemm.h), templated on the element type.
//
// Block tile 128 x 128 computed by NW wavefronts:
//   NW = 8 (default): 512 threads, 2 x 4 wavefronts, wave tile 64 x 32 held as 4 x 2 accumulators of a 16x16x4 MFMA
//           (fp64: 64 VGPRs), two workgroups per CU = FOUR wavefronts per SIMD;
//   NW = 4          : 256 threads, 2 x 2 wavefronts, wave tile 64 x 64 (4 x 4 accumulators, 128 VGPRs), two
//           workgroups per CU = two wavefronts per SIMD (the round-1 engine, kept for ablation: GPRHIP_ENG_WAVES=4).
// Why four wavefronts per SIMD: measured on MI355X (tools/gemm_check.hip PEAK=1) a register-only stream of
// independent v_mfma_f64_16x16x4_f64 reaches 78.0 TFLOP/s with two wavefronts per SIMD but only 35.8 with one --
// a single wavefront can feed the fp64 matrix pipe at half rate only.  With two resident wavefronts every stall
// of either one (fragment-read waits, barriers, prologue, epilogue) therefore idles the pipe at half weight and
// nothing can cover it; with four, any two ready wavefronts keep it full.
// A k-stage is 128 bytes of k per row (16 doubles / 32 floats); two stages are double-buffered so
// the global loads of stage t+1 are in flight while stage t is multiplied (one barrier per stage).
//
// LDS images (elements; identical byte geometry for both types):
//   x-major operand (global contiguous along k):  [128][BK+pad]  fp64 stride 18: the fragment read
//       (lane -> row l&15, k l>>4) hits 32 distinct 8-byte bank pairs per half-wave; fp32 stride 36
//       (16-byte aligned rows; 2-way conflicts on a read that is far off the critical path);
//   k-major operand (global contiguous along x):  [BK][144]  -- the k-rows a half-wave touches fall on
//       disjoint bank ranges.
// Fragment maps (cdna_hip_programming.md section 3): A lane l holds A[i=l&15][k=l>>4], B lane l holds
// B[k=l>>4][j=l&15]; C/D lane l reg r holds C[(l>>4)+4r][l&15] for f64 and C[4(l>>4)+r][l&15] for f32.
#include <type_traits>
#include <cstdlib>
#include "mfma_gemm.h"

namespace gprhip {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <typename T>
struct Geo;
template <>
struct Geo<double> {
  static constexpr int BKT = 16;   // k-depth of a stage
  static constexpr int EPV = 2;    // elements per 16-byte vector
  static constexpr int XS = 18;    // x-major row stride (elements)
  typedef d2 vec;
  typedef d4 acc_t;
  static __device__ __forceinline__ acc_t mfma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int crow(int lq, int r) { return lq + 4 * r; }
};
template <>
struct Geo<float> {
  static constexpr int BKT = 32;
  static constexpr int EPV = 4;
  static constexpr int XS = 36;
  typedef f4 vec;
  typedef f4 acc_t;
  static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int crow(int lq, int r) { return 4 * lq + r; }
};
constexpr int KS = 144;               // k-major row stride (elements)
constexpr int LDS_BYTES = 4 * 128 * 18 * 8;  // 2 stages x 2 operands x 18432 bytes

// Block -> (row tile, column tile, k-slice).  The hardware dispatcher places block b on XCD b % 8
// (observed, used for speed only): blocks are renumbered so that the blocks one XCD receives
// back-to-back share operand panels through that XCD's L2 --
//   full grids   : XCD x owns row panels x, x+8, ...; all column tiles of a panel run together
//                  (the A panel is fetched once per XCD instead of once per column tile);
//   split-K grids: XCD x owns k-slices x, x+8, ...; all output tiles of a slice run together
//                  (every tile of the slice streams the same rows of the operand).
template <typename T>
__device__ __forceinline__ void tile_of_block(const GemmArgsT<T>& g, int b, int nbm, int nbn, int& bm,
                                              int& bn, int& slice) {
  slice = 0;
  if (g.upper_only) {
    const int total = nbn * (nbn + 1) / 2;
    int t = b;
    if (g.kslices > 1) {
      if (g.kslices % 8 == 0) {
        const int x = b & 7, q = b >> 3;
        slice = (q / total) * 8 + x;
        t = q % total;
      } else {
        slice = b / total;
        t = b % total;
      }
    }
    // enumerate (bm <= bn) pairs; heavy tiles first for the triangular k-ranges
    if (g.tri == TRI_KHI_MIN) t = total - 1 - t;
    int c = 0;
    while (t >= c + 1) {
      t -= c + 1;
      ++c;
    }
    bn = c;
    bm = t;
    return;
  }
  if (g.kslices > 1) {  // full grid with split-K: the slices of one output tile set run back to back
    slice = b / (nbm * nbn);
    b %= nbm * nbn;
  }
  if (g.tri == TRI_KLO_BM) {
    bm = b / nbn;
    bn = b % nbn;
    return;
  }
  int bi;
  if (g.order == 1 && nbm % 8 == 0) {  // XCD-local row panels (all column tiles of a panel together)
    const int x = b & 7, q = b >> 3;
    bm = (q / nbn) * 8 + x;
    bi = q % nbn;
  } else {  // column-tile major: every resident block has the same k-range and shares the B panel
    bi = b / nbm;
    bm = b % nbm;
  }
  bn = (g.tri == TRI_KHI_BN) ? (nbn - 1 - bi) : bi;
}

// Value of lane (l ^ O) within each row of 16 lanes, O in {1, 2, 4, 8}, through DPP register moves (no LDS
// crossbar): quad_perm for 1 and 2, row_half_mirror + quad_perm[3,2,1,0] for 4 (l^7^3), row_ror:8 for 8.
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int O>
__device__ __forceinline__ double xor16(double v) {
  if constexpr (O == 1) return dpp_mov<0xB1>(v);
  else if constexpr (O == 2) return dpp_mov<0x4E>(v);
  else if constexpr (O == 4) return dpp_mov<0x1B>(dpp_mov<0x141>(v));
  else return dpp_mov<0x128>(v);
}

// CS: the launch carries weighted column sums of the raw A operand (GemmArgsT::cs_w) -- a separate
// instantiation, so that launches without them keep their inner loop unchanged
template <typename T, int OP, bool CS, int NW>
__global__ __launch_bounds__(NW * 64, NW / 2) void gemm_kernel(GemmArgsT<T> g) {
  constexpr int NT = NW * 64;     // threads per workgroup
  constexpr int WC = NW / 2;      // wave columns (wave rows: 2, 64 rows each)
  constexpr int NJ = 8 / WC;      // 16-column sub-tiles per wavefront
  constexpr int NP = 1024 / NT;   // staging passes per operand and stage (1024 16-byte vectors per operand tile)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  typedef Geo<T> G;
  typedef typename G::vec vec;
  typedef typename G::acc_t acc_t;
  constexpr int BK = G::BKT, EPV = G::EPV, XS = G::XS;
  constexpr int STAGE = 128 * XS;          // elements per operand per stage (18432 bytes)
  constexpr int VPR = 128 / EPV;           // 16-byte vectors per k-major row
  constexpr int KPP = NT / VPR;            // k-rows covered per staging pass
  constexpr bool A_KMAJ = (OP == OP_TN);
  constexpr bool B_XMAJ = (OP == OP_NT);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wr = wid / WC, wc = wid % WC;
  const int l15 = lane & 15, lq = lane >> 4;

  const int nbm = g.M / TILE, nbn = g.N / TILE;
  int bm, bn, slice;
  tile_of_block(g, blockIdx.x, nbm, nbn, bm, bn, slice);
  if (g.nbatch > 1) {  // kernel-argument copy: advance to this block's problem
    g.A += (int64_t)blockIdx.y * g.batch_a;
    g.B += (int64_t)blockIdx.y * g.batch_b;
    g.C += (int64_t)blockIdx.y * g.batch_c;
  }

  // k-range of this block, in elements
  int k_lo = 0, k_hi = g.K;
  if (g.kslices > 1) {
    // triangular operands: slices end on 128-block boundaries, so a diagonal block never straddles two slices
    const int gran = (g.tri != TRI_NONE) ? TILE : BK;
    int per = ((g.K / gran + g.kslices - 1) / g.kslices) * gran;
    k_lo = slice * per;
    k_hi = min(g.K, k_lo + per);
  }
  switch (g.tri) {
    case TRI_KHI_BN: k_hi = min(k_hi, (bn + 1) * TILE); break;
    case TRI_KLO_BN: k_lo = max(k_lo, bn * TILE); break;
    case TRI_KLO_BM: k_lo = max(k_lo, bm * TILE); break;
    case TRI_KLO_MAX: k_lo = max(k_lo, max(bm, bn) * TILE); break;
    case TRI_KHI_MIN: k_hi = min(k_hi, (min(bm, bn) + 1) * TILE); break;
    case TRI_BAND: k_lo = max(k_lo, bm * TILE); k_hi = min(k_hi, (bn + 1) * TILE); break;
    default: break;
  }
  const int nk = (k_hi - k_lo) / BK;

  acc_t acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (acc_t){0, 0, 0, 0};

  // ---- per-thread staging coordinates (16-byte vectors)
  // x-major tile: row = (tid>>3) + (NT/8)p, vector tid&7 of the row's 128 bytes;
  // k-major tile: k = tid/VPR + KPP*p, vector tid%VPR of the row's 128 elements
  const int xm_row = tid >> 3, xm_v = tid & 7;
  const int km_k = tid / VPR, km_v = tid % VPR;

  const T* Ag;
  int64_t a_step;  // global advance per k-stage
  if (A_KMAJ) {
    Ag = g.A + (int64_t)(k_lo + km_k) * g.lda + (int64_t)bm * TILE + EPV * km_v;
    a_step = (int64_t)BK * g.lda;
  } else {
    Ag = g.A + (int64_t)(bm * TILE + xm_row) * g.lda + k_lo + EPV * xm_v;
    a_step = BK;
  }
  const T* Bg;
  int64_t b_step;
  if (B_XMAJ) {
    Bg = g.B + (int64_t)(bn * TILE + xm_row) * g.ldb + k_lo + EPV * xm_v;
    b_step = BK;
  } else {
    Bg = g.B + (int64_t)(k_lo + km_k) * g.ldb + (int64_t)bn * TILE + EPV * km_v;
    b_step = (int64_t)BK * g.ldb;
  }
  // The per-k weights come through the scalar cache.  fp64: a wavefront stages one k-row per pass (km_k = tid >> 6
  // is wavefront-uniform).  fp32: a wavefront stages two adjacent k-rows (km_k = 2 * wave + half); both weights are
  // fetched with wavefront-uniform addresses and each half-wave selects its own.
  constexpr int RPW = 64 / VPR;  // k-rows per wavefront and pass: 1 (fp64) or 2 (fp32)
  const int km_ku = __builtin_amdgcn_readfirstlane(km_k);  // first k-row of this wavefront
  const bool hi_half = (RPW == 2) && (lane >= 32);
  const double* sk = (A_KMAJ && g.scale_k) ? g.scale_k + k_lo + km_ku : nullptr;
  auto weight = [&](const double* w, int off) -> double {
    if (RPW == 1) return w[off];
    const double w0 = w[off], w1 = w[off + 1];
    return hi_half ? w1 : w0;
  };

  // the per-k weights of the A operand are fetched with the tile and applied when the tile is
  // written to LDS, so the multiply never waits on a load that was just issued
  T rs[NP];
#pragma unroll
  for (int q = 0; q < NP; ++q) rs[q] = (T)1;
  // weighted column sums of raw A (cs_*): only the diagonal tile of each k-slice carries them
  const bool do_cs = CS && A_KMAJ && bm == bn;
  const double* cw = do_cs ? g.cs_w + k_lo + km_ku : nullptr;
  double cwv[NP];
#pragma unroll
  for (int q = 0; q < NP; ++q) cwv[q] = 0.0;
  double csum[EPV];
#pragma unroll
  for (int e = 0; e < EPV; ++e) csum[e] = 0.0;
  auto load_global = [&](int t, vec (&ra)[NP], vec (&rb)[NP]) {
    if (g.lab_noadvance) t = 0;  // engine lab only: every stage re-reads the first one (cache-resident operands)
    const T* ap = Ag + (int64_t)t * a_step;
    const T* bp = Bg + (int64_t)t * b_step;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      if (A_KMAJ) {
        ra[p] = *reinterpret_cast<const vec*>(ap + (int64_t)(KPP * p) * g.lda);
        if (sk) rs[p] = (T)weight(sk, t * BK + KPP * p);
        if (do_cs) cwv[p] = weight(cw, t * BK + KPP * p);
      } else {
        ra[p] = *reinterpret_cast<const vec*>(ap + (int64_t)((NT / 8) * p) * g.lda);
      }
      if (B_XMAJ) {
        rb[p] = *reinterpret_cast<const vec*>(bp + (int64_t)((NT / 8) * p) * g.ldb);
      } else {
        rb[p] = *reinterpret_cast<const vec*>(bp + (int64_t)(KPP * p) * g.ldb);
      }
    }
  };
  auto store_lds = [&](int stage, const vec (&ra)[NP], const vec (&rb)[NP]) {
    T* As = smem + stage * 2 * STAGE;
    T* Bs = As + STAGE;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      if (A_KMAJ) {
        vec v = ra[p];
        if (do_cs) {
#pragma unroll
          for (int e = 0; e < EPV; ++e) csum[e] += (double)v[e] * cwv[p];
        }
        if (sk) v *= rs[p];
        *reinterpret_cast<vec*>(As + (km_k + KPP * p) * KS + EPV * km_v) = v;
      } else
        *reinterpret_cast<vec*>(As + (xm_row + (NT / 8) * p) * XS + EPV * xm_v) = ra[p];
      if (B_XMAJ)
        *reinterpret_cast<vec*>(Bs + (xm_row + (NT / 8) * p) * XS + EPV * xm_v) = rb[p];
      else
        *reinterpret_cast<vec*>(Bs + (km_k + KPP * p) * KS + EPV * km_v) = rb[p];
    }
  };

  // fragment base offsets (elements) inside a stage
  const int a_frag = A_KMAJ ? (lq * KS + wr * 64 + l15) : ((wr * 64 + l15) * XS + lq);
  // the 8 column sub-tiles (16 columns each) of the block tile are dealt round-robin to the WC wave
  // columns (wave column wc owns sub-tiles wc, wc+WC, ...): inside the diagonal block of a
  // triangular operand the live sub-tiles then split evenly between the waves of a row
  const int b_frag = B_XMAJ ? ((wc * 16 + l15) * XS + lq) : (lq * KS + wc * 16 + l15);
  constexpr int A_TM = A_KMAJ ? 16 : 16 * XS;  // advance per 16-row sub-tile
  constexpr int A_KK = A_KMAJ ? 4 * KS : 4;    // advance per k-step of 4
  constexpr int B_TN = B_XMAJ ? 16 * WC * XS : 16 * WC;  // advance per owned column sub-tile (every WC-th one)
  constexpr int B_KK = B_XMAJ ? 4 : 4 * KS;

  // Inside the diagonal 128-block of a triangular B operand, column sub-tile cj (16 columns,
  // cj = 0..7 across the block tile) only meets non-zeros in k-stages that start at or before its last
  // column (k-range ends at the diagonal, TRI_KHI_BN) or end at or after its first column (k-range
  // starts at it, TRI_KLO_BN): the MFMAs of the other
  // (sub-tile, stage) pairs multiply zeros and are skipped -- about half of that block's work.
  const int diag_first = (g.tri == TRI_KHI_BN && k_hi == (bn + 1) * TILE) ? nk - TILE / BK
                         : (g.tri == TRI_KLO_BN && k_lo == bn * TILE) ? 0 : -(1 << 30);
  const int cj0 = wc;  // first owned column sub-tile; sub-tile j of this wave is WC*j + wc
  // Diagonal tile of an upper_only (SYRK-shaped) launch: only the 16x16 sub-tiles on or above the diagonal are
  // consumed (row sub-tile 4*wr + i <= column sub-tile wc + WC*j); the others are not computed.  Bit i*4+j.
  constexpr unsigned ALL_LIVE = NJ == 4 ? 0xFFFFu : 0x3333u;
  unsigned sub_live = ALL_LIVE;
  // (not in the column-sum variant: measured slower there, its diagonal blocks already carry the extra sums)
  if (!CS && g.upper_only && bm == bn) {
    sub_live = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        if (4 * wr + i <= wc + WC * j) sub_live |= 1u << (i * 4 + j);
  }
  // live column sub-tiles [jlo, jhi] of k-stage t (see above); all 8 outside the diagonal block of a triangular B
  auto live_range = [&](int t, int& jlo, int& jhi) {
    jlo = 0;
    jhi = 7;
    const int u = t - diag_first;
    if (u >= 0 && u < TILE / BK) {  // stage u of the diagonal block covers k in [BK*u, BK*u + BK)
      if (g.tri == TRI_KHI_BN) jlo = (BK * u) / 16; else jhi = (BK * u + BK - 1) / 16;
    }
  };
  auto read_frags = [&](int stage, int kk, T (&af)[4], T (&bf)[NJ]) {
    const T* As = smem + stage * 2 * STAGE;
    const T* Bs = As + STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = As[a_frag + i * A_TM + kk * A_KK];
#pragma unroll
    for (int j = 0; j < NJ; ++j) bf[j] = Bs[b_frag + j * B_TN + kk * B_KK];
  };
  auto mma = [&](const T (&af)[4], const T (&bf)[NJ], int jlo, int jhi, bool all) {
    if (all) {  // every sub-tile live: the common case, one straight MFMA stream
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = G::mfma(af[i], bf[j], acc[i][j]);
      return;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (cj0 + WC * j >= jlo && cj0 + WC * j <= jhi) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (sub_live >> (i * 4 + j) & 1u) acc[i][j] = G::mfma(af[i], bf[j], acc[i][j]);
      }
    }
  };

  if (nk > 0) {
    // Software pipeline of one k-stage (NKK k-steps of 4; fragments double-buffered in registers):
    //   top        : global loads of stage t+1 -> registers
    //   k-step kk  : LDS reads of the fragments of k-step kk+1, then the MFMAs of k-step kk
    //   k-step KST : ... the LDS refill of the other stage buffer (stage t+1) goes out before this step's MFMAs and
    //                drains under them
    //   last k-step: barrier (refill complete, every wave done reading this stage), LDS reads of the first fragments
    //                of stage t+1, then the last MFMAs of stage t
    // so a wavefront never waits on an LDS read it has just issued, and the barrier has a full k-step of matrix
    // work behind it.  This matters because the co-resident wavefronts of a SIMD run the same code in step: their
    // waits coincide, and the fp64 matrix pipe needs two issuing wavefronts to stay full (tools/gemm_check PEAK=1).
    constexpr int NKK = BK / 4;
    constexpr int KST = NKK - 2;
    vec ra[NP], rb[NP];
    load_global(0, ra, rb);
    store_lds(0, ra, rb);
    __syncthreads();
    T af[2][4], bf[2][NJ];
    read_frags(0, 0, af[0], bf[0]);
    for (int t = 0; t < nk; ++t) {
      const bool more = (t + 1 < nk);
      if (more) load_global(t + 1, ra, rb);
      int jlo, jhi;
      live_range(t, jlo, jhi);
      const bool any = !(jlo > cj0 + WC * (NJ - 1) || jhi < cj0);
      const bool all = jlo <= cj0 && jhi >= cj0 + WC * (NJ - 1) && sub_live == ALL_LIVE;
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) {
        if (kk + 1 < NKK) read_frags(t & 1, kk + 1, af[(kk + 1) & 1], bf[(kk + 1) & 1]);
        if (kk == KST && more) store_lds((t + 1) & 1, ra, rb);
        if (kk == NKK - 1) {
          __syncthreads();
          if (more) read_frags((t + 1) & 1, 0, af[0], bf[0]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (any) mma(af[kk & 1], bf[kk & 1], jlo, jhi, all);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  if (do_cs) {
    // threads with the same km_v hold partial sums of the same EPV columns for different k-rows: combine the
    // NT / VPR groups through LDS (free after the main loop's final barrier)
    double* red = reinterpret_cast<double*>(smem_raw);
#pragma unroll
    for (int e = 0; e < EPV; ++e) red[km_k * TILE + EPV * km_v + e] = csum[e];
    __syncthreads();
    if (tid < TILE) {
      double s = 0.0;
#pragma unroll
      for (int q = 0; q < NT / VPR; ++q) s += red[q * TILE + tid];
      g.cs_out[(int64_t)slice * g.N + bm * TILE + tid] = s;
    }
  }

  // ---- epilogue
  const int rowb = bm * TILE + wr * 64, col0 = bn * TILE + wc * 16 + l15;  // + CST per owned sub-tile
  constexpr int CST = 16 * WC;
  T* Cp = g.C + (int64_t)slice * g.slice_stride + col0;
  const T alpha = (T)g.alpha, beta = (T)g.beta;
  if (g.epi_rows_a) {
    // fused X~ epilogue: C[i][j] = ra[i]*acc - rb[i]*M[i][j] - rc[i]*cv[j]
    //   (X~ = diag(is) Q' R~^-T - diag(v) V - w t~^T of the gradient pass)
    const T* Mp = g.epi_mat + col0;
    T cv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) cv[j] = (T)g.epi_col[col0 + j * CST];
    // per 16-row group: issue every load of the group (4 rows x (4 M values + 3 row scalars)) before the first
    // use, so a block pays 4 memory round trips here instead of one per row
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      T mv[4][NJ], ra[4], rb[4], rc[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rowb + i * 16 + G::crow(lq, r);
        ra[r] = (T)g.epi_rows_a[row];
        rb[r] = (T)g.epi_rows_b[row];
        rc[r] = (T)g.epi_rows_c[row];
#pragma unroll
        for (int j = 0; j < NJ; ++j) mv[r][j] = Mp[(int64_t)row * g.epi_ldm + j * CST];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rowb + i * 16 + G::crow(lq, r);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          Cp[(int64_t)row * g.ldc + j * CST] = ra[r] * acc[i][j][r] - rb[r] * mv[r][j] - rc[r] * cv[j];
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    T old[NJ][4];
    if (beta != (T)0) {  // all loads of the row group in flight before the first use
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) old[j][r] = Cp[(int64_t)(rowb + i * 16 + G::crow(lq, r)) * g.ldc + j * CST];
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        T v = alpha * acc[i][j][r];
        if (beta != (T)0) v += beta * old[j][r];
        Cp[(int64_t)(rowb + i * 16 + G::crow(lq, r)) * g.ldc + j * CST] = v;
        acc[i][j][r] = v;  // the stored value, for the row reductions below
      }
  }
  if (g.rp_sumsq) {
    // wavefront row reductions over this wave's 16*NJ columns: NJ values in-thread, then the 16 lanes of a
    // row group by xor-shuffles; lane l15 == 0 writes one partial per (row, tile, wave column)
    double bv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) bv[j] = g.rp_dot ? g.rp_vec[col0 + j * CST] : 0.0;
    double s2[16], sd[16];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double x2 = 0.0, xd = 0.0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const double v = (double)acc[i][j][r];
          x2 += v * v;
          xd += v * bv[j];
        }
        s2[i * 4 + r] = x2;
        sd[i * 4 + r] = xd;
      }
    // transposing butterfly over the 16 lanes of a row group: 8+4+2+1 exchanges leave lane l15 with the
    // total of value k = l15 (i = l15 >> 2, r = l15 & 3)
    auto fold = [&](auto oc) {
      constexpr int o = decltype(oc)::value;
      const bool hi = (l15 & o) != 0;
#pragma unroll
      for (int k = 0; k < o; ++k) {
        const double keep = hi ? s2[k + o] : s2[k];
        const double send = hi ? s2[k] : s2[k + o];
        s2[k] = keep + xor16<o>(send);
        if (g.rp_dot) {
          const double keepd = hi ? sd[k + o] : sd[k];
          const double sendd = hi ? sd[k] : sd[k + o];
          sd[k] = keepd + xor16<o>(sendd);
        }
      }
    };
    fold(std::integral_constant<int, 8>{});
    fold(std::integral_constant<int, 4>{});
    fold(std::integral_constant<int, 2>{});
    fold(std::integral_constant<int, 1>{});
    const int64_t idx = (int64_t)(WC * bn + wc) * g.M + (rowb + (l15 >> 2) * 16 + G::crow(lq, l15 & 3));
    g.rp_sumsq[idx] = s2[0];
    if (g.rp_dot) g.rp_dot[idx] = sd[0];
  }
}

static int g_waves = 8;  // wavefronts per workgroup of the engine (8, or 4 for the round-1 geometry)

int gemm_row_parts_per_tile() { return g_waves / 2; }

template <typename T, int NW>
static void set_lds_limit() {
  const void* ks[] = {reinterpret_cast<const void*>(&gemm_kernel<T, OP_NN, false, NW>),
                      reinterpret_cast<const void*>(&gemm_kernel<T, OP_NT, false, NW>),
                      reinterpret_cast<const void*>(&gemm_kernel<T, OP_TN, false, NW>),
                      reinterpret_cast<const void*>(&gemm_kernel<T, OP_TN, true, NW>)};
  for (const void* k : ks)
    GPR_HIP(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
}

void gemm_init() {
  static bool done = false;
  if (done) return;
  if (const char* e = getenv("GPRHIP_ENG_WAVES")) g_waves = (atoi(e) == 4) ? 4 : 8;  // ablation only, read once
  set_lds_limit<double, 4>();
  set_lds_limit<double, 8>();
  set_lds_limit<float, 4>();
  set_lds_limit<float, 8>();
  done = true;
}

template <typename T, int NW>
static void launch_gemm_nw(GemmOp op, const GemmArgsT<T>& g, dim3 grid, hipStream_t stream) {
  dim3 block(NW * 64);
  switch (op) {
    case OP_NN: hipLaunchKernelGGL((gemm_kernel<T, OP_NN, false, NW>), grid, block, LDS_BYTES, stream, g); break;
    case OP_NT: hipLaunchKernelGGL((gemm_kernel<T, OP_NT, false, NW>), grid, block, LDS_BYTES, stream, g); break;
    case OP_TN:
      if (g.cs_w) hipLaunchKernelGGL((gemm_kernel<T, OP_TN, true, NW>), grid, block, LDS_BYTES, stream, g);
      else hipLaunchKernelGGL((gemm_kernel<T, OP_TN, false, NW>), grid, block, LDS_BYTES, stream, g);
      break;
  }
}

template <typename T>
static void launch_gemm_t(GemmOp op, const GemmArgsT<T>& g, hipStream_t stream) {
  constexpr int BK = Geo<T>::BKT;
  if (g.M % TILE || g.N % TILE || g.K % BK || g.M <= 0 || g.N <= 0 || g.K <= 0) {
    set_error("gprhip: launch_gemm: dimensions must be padded (M,N % 128, K % stage depth)");
    throw HipFail{ST_BAD_ARG};
  }
  const int nbm = g.M / TILE, nbn = g.N / TILE;
  int tiles = g.upper_only ? nbn * (nbn + 1) / 2 : nbm * nbn;
  if (g.cs_w && (op != OP_TN || !g.upper_only || !g.cs_out)) {
    set_error("gprhip: launch_gemm: column sums ride on upper_only TN launches");
    throw HipFail{ST_BAD_ARG};
  }
  if (g.kslices > 1 && (g.beta != 0.0 || g.epi_rows_a || g.rp_sumsq)) {
    set_error("gprhip: launch_gemm: split-K launches write plain partial products (no beta / fused epilogue)");
    throw HipFail{ST_BAD_ARG};
  }
  dim3 grid(tiles * (g.kslices > 1 ? g.kslices : 1), g.nbatch > 1 ? g.nbatch : 1);
  if (g_waves == 8) launch_gemm_nw<T, 8>(op, g, grid, stream);
  else launch_gemm_nw<T, 4>(op, g, grid, stream);
  GPR_HIP(hipGetLastError());
}

void launch_gemm(GemmOp op, const GemmArgs& g, hipStream_t stream) { launch_gemm_t<double>(op, g, stream); }
void launch_gemm(GemmOp op, const GemmArgsF& g, hipStream_t stream) { launch_gemm_t<float>(op, g, stream); }

}  // namespace gprhip
