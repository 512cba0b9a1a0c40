// Row passes and finish stage of problems with 65 .. 128 inducing points (one 128-column tile; d <= 16 point dimensions,
// 1 + d + D <= 32 with a projection from D input dimensions, no multiscales, fp64, up to 2^22 rows per shard) and, in the
// two-tile form further down (MidGeo<256>), 129 .. 256 (up to 32768 rows per shard: mid_path_fits) -- the regime the
// reference's own default lands in: Optim.get_kernel_inducing takes min (n_inputs / 10) 1000 inducing points
// (lib/fitc_gp.ml:1474-1479), i.e. 65 .. 128 of them for every data set of 650 .. 1280 points.  Through the engine such an
// evaluation is ~30 dependent launches of 5-35 us each (n = 2000, m = 128: 0.38 ms, profiles/r05_latency.txt): six
// contraction launches whose 128 x 128 x 128 tiles take 14 us of MFMAs on one CU however few rows there are, their row
// kernels and reductions, and eight launches of m x m work.  Here, as in small.hip for m <= 64, each pass is ONE kernel
// per 64-row block plus one fixed-order reduction of the per-workgroup partial sums into the exchange buffers (same
// buffers, same layout as do_pass1 / do_pass2 write: everything around the passes, sharded evaluations included, is shared):
//   pass 1: K (lib/cov_se_iso.ml:128-159, lib/cov_se_fat.ml:224-240), V = K U^-1 (lib/fitc_gp.ml:226-227), r, s, 1/s
//           (:155-166, :222-223), B~ part = V^T diag(is) V, c~ part = V^T (is y)
//   pass 2: Q' = V R~^-1, q_diag, w, v (:1048, :1092-1108, :1158-1181), X~ = diag(is) Q' R~^-T - diag(v) V - w t~^T,
//           X = X~ U^-T (:931-939, :1204-1206), E = X .* K and its moments against the points (:975-1003),
//           G~ part = V^T diag(v) V (:1198-1203)
//   finish: B~^-1, W~, W = U^-1 W~ U^-T, the trace terms against K_m (:956-973, lib/utils.ml:196-220) in one workgroup
// What differs from small.hip: a 128 x 128 triangular inverse is 128 KB -- the LDS (160 KB) holds the block's own tiles
// (V and Q'/X~/X/E: 2 x 66 KB) and NOTHING else, so the triangular operands U^-1 and R~^-1 never enter it: every wavefront
// fetches the B fragments of its MFMAs straight from memory (the matrices are L2-resident: 128 KB each, read by every
// workgroup), one k-batch ahead of the MFMAs that use them.  A wavefront owns COLUMN tiles here (16-column tiles w and
// 7 - w: for an upper-triangular operand the k-ranges of tile j are 16 (j + 1) rows deep, so the pair is 9 batches of 16
// for every wavefront), all 64 rows of the block; the two Gram accumulations give wavefront w the tile rows w and 7 - w of
// the upper triangle (8 - w and w + 1 tiles: 9 each).  Everything that depends on w is a template parameter, so that all
// fragment and accumulator indices are compile-time register names.
#include <algorithm>
#include <type_traits>

#include "kernels.h"
#include "exp_fast.h"

namespace gprhip {

namespace {

// Geometry by padded inducing-point count MPV (128: m <= 128; 256: 129 .. 256, "two tiles"): the row block shrinks with the
// tile width so that a block's tile stays 64 KB; the two-tile form leaves both Gram accumulations to the engine's SYRK-shaped
// launches (136 upper 16 x 16 tiles are 272 accumulator registers per wavefront, and 280 KB of partial sums per workgroup).
template <int MPV>
struct MidGeo {
  static constexpr int MP = MPV;          // padded inducing points
  static constexpr int MLD = MPV + 2;     // leading dimension of the LDS tiles (row stride = 4 dwords mod 64: the half-wave
                                          //   fragment reads fall on 64 different banks, as SLD = 66 in small.hip)
  static constexpr int MRB = MPV == 128 ? 64 : 32;  // training points per block
  static constexpr int MCT = MPV / 16;    // 16-column tiles
  static constexpr int NJ = MCT / 4;      // ... per wavefront
  static constexpr int RT = MRB / 16;     // 16-row tiles of a block
  static constexpr bool GRAM = MPV == 128;
};
// 16-column tile jj (ascending) of wavefront W: {W, 7 - W} or {W, 7 - W, 8 + W, 15 - W} -- for an upper-triangular operand
// tile j is 16 (j + 1) rows deep, so every wavefront gets the same number of k-batches (9 / 34)
template <int MPV, int W>
__host__ __device__ constexpr int wave_tile(int jj) {
  return jj == 0 ? W : jj == 1 ? 7 - W : jj == 2 ? 8 + W : 15 - W;
}
__device__ __forceinline__ int wave_tile_rt(int mpv, int wv, int jj) {
  return jj == 0 ? wv : jj == 1 ? 7 - wv : jj == 2 ? 8 + wv : 15 - wv;
}

constexpr int MP = 128;        // the one-tile geometry, spelled out for the Gram code (which exists for it only)
constexpr int MLD = 130;
constexpr int MRB = 64;
constexpr int MCT = 8;
constexpr int MNTU = 36;       // upper 16 x 16 tiles of a 128 x 128 symmetric accumulation
constexpr int MGLEN = MNTU * 256;
constexpr int M1LEN = MGLEN + MP + 4;  // pass-1 partial: B~ upper tiles | c~ part | sum log s, sum y^2/s, sum r/s, -
// pass-2 partial: G~ upper tiles | moment rows (1, p_k (d), x_big (D)) x 128 | 8 scalars
__host__ __device__ constexpr int m2len(int d, int D) { return MGLEN + (1 + d + D) * MP + 8; }
// two tiles: pass-1 partial = the four scalars, pass-2 partial = moment rows x 256 | 8 scalars
constexpr int W1LEN = 4;
__host__ __device__ constexpr int w2len(int d, int D) { return (1 + d + D) * 256 + 8; }
// index of upper tile (it <= jt), row-major over the upper triangle
__host__ __device__ constexpr int tix(int it, int jt) { return it * MCT - it * (it - 1) / 2 + (jt - it); }

typedef double sd4 __attribute__((ext_vector_type(4)));
// A pointer into device memory that says so: the triangular operands are fetched through pointers laundered once per block
// (see mid_pass1_body), and a laundered pointer of plain type is a generic one -- its loads become flat_load, which may hit
// the LDS as far as the hardware knows, so every use waits for ALL outstanding memory AND LDS operations
// (s_waitcnt vmcnt(0) lgkmcnt(0) in front of every k-batch: no load ever overlapped an MFMA).
typedef const __attribute__((address_space(1))) double* gmem_ptr;
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

// acc[i][jj] = rows 16 i .. 16 i + 15 of A (LDS, [MRB][MLD]) times the columns of this wavefront's tile jj of the
// upper-triangular B (memory, row-major [MP][MP], exact zeros below the diagonal): tile j needs k < 16 (j + 1) only, i.e.
// the k-batches kb <= j.  The batch loop stays ROLLED, its operand pointers stepping by one batch: fully unrolled, each of
// the fragment loads gets an address register of its own (a k-step is 1-2 KB of B away from the next, beyond the
// instruction's immediate offset), all of them loop-invariant and therefore computed in front of the block loop and
// spilled.  Fragments of the batch in hand and of the two behind it are in registers: a batch is 16-32 MFMAs (1000-2000
// cycles), one of them does not cover a round trip to the L2.
// Three fragment buffers used in place, the loop unrolled by three: the loads of batch kb + 2 go out before the MFMAs of
// batch kb, and nothing ever copies a buffer (a register move would have to wait for the load it moves -- with two
// buffers rotated by moves every batch waited for the loads issued at its own top: a memory round trip per batch, 25 of
// pass 2's 49 us at n = 2000, m = 128).  Every stage issues the same number of loads (row / column indices clamped into the
// matrix where a tile no longer needs them) so that the waits in front of the MFMAs count exactly.
template <int MPV, int W>
__device__ __forceinline__ void tri_nn(const double* A, gmem_ptr B, int l15, int lq,
                                       sd4 (&acc)[MidGeo<MPV>::RT][MidGeo<MPV>::NJ]) {
  using G = MidGeo<MPV>;
  constexpr int NJ = G::NJ, RT = G::RT, LDA = G::MLD, LDB = G::MP;
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) acc[i][jj] = sd4{0.0, 0.0, 0.0, 0.0};
  const gmem_ptr bq = B + lq * LDB + l15;        // B[16 kb + 4 s + lq][16 J + l15] at bq + (16 kb + 4 s) LDB + 16 J
  const double* ap = A + l15 * LDA + lq;         // A[16 i + l15][16 kb + 4 s + lq] at ap + 16 i LDA + 4 s, ap stepping by 16
  constexpr int JL = wave_tile<MPV, W>(NJ - 1);
  double buf[3][4][NJ];
  auto fetch = [&](int kb, double (&dst)[4][NJ]) {
    const gmem_ptr bk = bq + (int64_t)16 * min(kb, G::MCT - 1) * LDB;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) dst[s][jj] = bk[4 * s * LDB + 16 * wave_tile<MPV, W>(jj)];
  };
  auto work = [&](int kb, const double (&src)[4][NJ]) {
    double af[4][RT];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < RT; ++i) af[s][i] = ap[16 * i * LDA + 4 * s];
    ap += 16;
#pragma unroll
    for (int jj = NJ - 1; jj >= 0; --jj)
      if (kb <= wave_tile<MPV, W>(jj)) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < RT; ++i) acc[i][jj] = mfma_f64(af[s][i], src[s][jj], acc[i][jj]);
      }
  };
  fetch(0, buf[0]);
  fetch(1, buf[1]);
#pragma unroll 1
  for (int kb = 0; kb <= JL; kb += 3) {
    fetch(kb + 2, buf[2]);
    work(kb, buf[0]);
    if (kb + 1 > JL) break;
    fetch(kb + 3, buf[0]);
    work(kb + 1, buf[1]);
    if (kb + 2 > JL) break;
    fetch(kb + 4, buf[1]);
    work(kb + 2, buf[2]);
  }
}

// ... times B^T, given Bt = B^T in memory: out[r][c] = sum_k A[r][k] Bt[k][c], B upper triangular: tile j needs k >= 16 j, the
// batches kb >= j.  (Fetched from B itself a fragment load touches 16 rows of 32 bytes each -- sixteen cache lines per
// instruction, a quarter of each used -- and the texture path, not the matrix pipe, sets the pace: the two-tile pass 2 took
// 80 us per block for 22 us of MFMAs.  From the transposed copy it is four full lines, as in tri_nn.)
template <int MPV, int W>
__device__ __forceinline__ void tri_nt(const double* A, gmem_ptr Bt, int l15, int lq,
                                       sd4 (&acc)[MidGeo<MPV>::RT][MidGeo<MPV>::NJ]) {
  using G = MidGeo<MPV>;
  constexpr int NJ = G::NJ, RT = G::RT, LDA = G::MLD, LDB = G::MP, J0 = wave_tile<MPV, W>(0);
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) acc[i][jj] = sd4{0.0, 0.0, 0.0, 0.0};
  const gmem_ptr bq = Bt + lq * LDB + l15;           // Bt[16 kb + 4 s + lq][16 J + l15] at bq + (16 kb + 4 s) LDB + 16 J
  const double* ap = A + l15 * LDA + lq + 16 * J0;
  double buf[3][4][NJ];
  auto fetch = [&](int kb, double (&dst)[4][NJ]) {
    const gmem_ptr bk = bq + (int64_t)16 * min(kb, G::MCT - 1) * LDB;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) dst[s][jj] = bk[4 * s * LDB + 16 * wave_tile<MPV, W>(jj)];
  };
  auto work = [&](int kb, const double (&src)[4][NJ]) {
    double af[4][RT];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < RT; ++i) af[s][i] = ap[16 * i * LDA + 4 * s];
    ap += 16;
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj)
      if (kb >= wave_tile<MPV, W>(jj)) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < RT; ++i) acc[i][jj] = mfma_f64(af[s][i], src[s][jj], acc[i][jj]);
      }
  };
  fetch(J0, buf[0]);
  fetch(J0 + 1, buf[1]);
#pragma unroll 1
  for (int kb = J0; kb < G::MCT; kb += 3) {
    fetch(kb + 2, buf[2]);
    work(kb, buf[0]);
    if (kb + 1 >= G::MCT) break;
    fetch(kb + 3, buf[0]);
    work(kb + 1, buf[1]);
    if (kb + 2 >= G::MCT) break;
    fetch(kb + 4, buf[1]);
    work(kb + 2, buf[2]);
  }
}

// acc[q] += (T^T diag(wt) T) upper tiles of this wavefront over the 64 rows of T (LDS): tile rows I0 = W (q = 0 .. 7 - W:
// column tiles W .. 7) and I1 = 7 - W (q = 8 - W .. 8: column tiles 7 - W .. 7)
template <int W>
__device__ __forceinline__ void gram_update(const double* T, const double* wt, int l15, int lq, sd4 (&acc)[9]) {
  constexpr int I0 = W, I1 = 7 - W;
  // (one lane-dependent base per array and compile-time offsets from it: written as T[k * MLD + ...] with k = 4 (..) + lq,
  //  the compiler turns the sum into an `or`, cannot pull the constant through the multiplication any more, and keeps one
  //  address register per k -- hundreds of them over this file, all loop-invariant, all spilled)
  const double* tq = T + lq * MLD + l15;
  const double* wq = wt + lq;
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    double a0[4], a1[4], bf[4][MCT];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int k4 = 4 * (4 * h + s);
      const double wk = wq[k4];
      a0[s] = tq[k4 * MLD + 16 * I0] * wk;
      a1[s] = tq[k4 * MLD + 16 * I1] * wk;
#pragma unroll
      for (int c = I0; c < MCT; ++c) bf[s][c] = tq[k4 * MLD + 16 * c];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int c = I0; c < MCT; ++c) acc[c - I0] = mfma_f64(a0[s], bf[s][c], acc[c - I0]);
#pragma unroll
      for (int c = I1; c < MCT; ++c) acc[(MCT - I0) + (c - I1)] = mfma_f64(a1[s], bf[s][c], acc[(MCT - I0) + (c - I1)]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// this wavefront's nine Gram tiles -> the partial buffer, tile tix(it, jt) as 16 x 16 row-major
template <int W>
__device__ __forceinline__ void store_gram(double* part, const sd4 (&acc)[9], int l15, int lq) {
  constexpr int I0 = W, I1 = 7 - W;
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    const int it = q < MCT - I0 ? I0 : I1;
    const int jt = q < MCT - I0 ? I0 + q : I1 + (q - (MCT - I0));
#pragma unroll
    for (int r = 0; r < 4; ++r) (part + lq * 16 + l15)[tix(it, jt) * 256 + 64 * r] = acc[q][r];
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

}  // namespace

// workgroups of a pass at most (each walks blocks b, b + groups, ...): one per CU (pass 2 fills the LDS; the per-workgroup
// partials of the one-tile form are 75 KB each)
constexpr int MID_GROUPS = 256;
int64_t mid_part_len(int mp, int d, int D) {
  return (int64_t)MID_GROUPS * (mp == 128 ? std::max(M1LEN, m2len(d, D)) : std::max(W1LEN, w2len(d, D)));
}
template <int MPV>
static int mid_groups(int rows_p) { return std::min(MID_GROUPS, rows_p / MidGeo<MPV>::MRB); }

// ---------------------------------------------------------------------------------------------------------------- pass 1
template <int MPV, int DT, int W>
__device__ __forceinline__ void mid_pass1_body(const MidPass1Args& a, double* lds) {
  using G = MidGeo<MPV>;
  constexpr int MP = G::MP, MLD = G::MLD, MRB = G::MRB, NJ = G::NJ, RT = G::RT;
  constexpr bool GRAM = G::GRAM;
  double* const Kt = lds;              // [MRB][MLD] K of the block, then V in place
  double* const xs = Kt + MRB * MLD;   // [MRB][DT]
  double* const isr = xs + MRB * DT;   // [MRB] 1/s
  double* const yisr = isr + MRB;      // [MRB] y/s
  double* const rsp = yisr + MRB;      // [4][MRB] row sums of V.^2 over each wavefront's columns
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lq = lane >> 4;
  const ExpK ek = exp_consts();
  const int col = tid % MP, rg = tid / MP;  // covariance / column-sum phases: thread = (column, 32 of the block's rows)
  const bool live_c = col < a.m;
  double z[DT];
#pragma unroll
  for (int k = 0; k < DT; ++k) z[k] = (k < a.d && live_c) ? a.Z[(int64_t)col * a.d + k] : 0.0;
  sd4 accB[GRAM ? 9 : 1];
#pragma unroll
  for (int q = 0; q < (GRAM ? 9 : 1); ++q) accB[q] = sd4{0.0, 0.0, 0.0, 0.0};
  double csum = 0.0, p_log = 0.0, p_y2 = 0.0, p_isr = 0.0;
  const int nblk = a.rows_p / MRB;
  for (int b = blockIdx.x; b < nblk; b += gridDim.x) {
    const int r0 = b * MRB;
    __syncthreads();
    for (int idx = tid; idx < MRB * DT; idx += 256) {
      const int r = idx / DT, k = idx % DT;
      xs[idx] = (k < a.d && r0 + r < a.rows) ? a.pts[(int64_t)(r0 + r) * a.d + k] : 0.0;
    }
    const double yreg = (tid < MRB && a.y && r0 + tid < a.rows) ? a.y[r0 + tid] : 0.0;  // used by the row phase below
    __syncthreads();
    {
      const double* xq = xs + rg * 32 * DT;
      double* kq = Kt + rg * 32 * MLD + col;
      const int rlive = a.rows - r0 - rg * 32;  // rows i < rlive of this thread's 32 are real
#pragma unroll 4
      for (int i = 0; i < 32; ++i) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < DT; ++k) {  // (dimensions beyond d are zero on both sides: they add exactly 0)
          const double diff = xq[i * DT + k] - z[k];
          acc = acc + diff * diff;
        }
        kq[i * MLD] = (i < rlive && live_c) ? exp_fast(a.cp.log_sf2 + a.cp.inv_ell2_05 * acc, ek) : 0.0;
      }
    }
    __syncthreads();
    sd4 acc[RT][NJ];
    // (the operand pointer is laundered per block: left loop-invariant, every B fragment of the product -- 72 loads -- is
    //  hoisted in front of the block loop and held in registers, in pass 2 -- three products -- spilled to scratch)
    gmem_ptr uinv = (gmem_ptr)a.uinv;
    asm volatile("" : "+s"(uinv));
    tri_nn<MPV, W>(Kt, uinv, l15, lq, acc);  // V = K U^-1
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double s = 0.0;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) s += acc[i][jj][r] * acc[i][jj][r];
        s = sum16(s);
        if (l15 == 0) (rsp + lq)[W * MRB + 16 * i + 4 * r] = s;
      }
    __syncthreads();  // every wavefront is done reading K
    {
      double* kq = Kt + lq * MLD + l15;                         // element r of acc[i][jj] is row 16 i + lq + 4 r,
      double* vq = a.V + (int64_t)(r0 + lq) * MP + l15;         //   column 16 J_jj + l15
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj) {
            if (GRAM) kq[(16 * i + 4 * r) * MLD + 16 * wave_tile<MPV, W>(jj)] = acc[i][jj][r];
            vq[(16 * i + 4 * r) * MP + 16 * wave_tile<MPV, W>(jj)] = acc[i][jj][r];
          }
    }
    if (tid < MRB) {  // r, s = r + sigma2, 1/s, sum log s  (as pass1_rows_kernel)
      const int row = r0 + tid;
      double rr = 0.0, is = 0.0, yis = 0.0;
      if (row < a.rows) {
        rr = a.cp.sf2 - ((rsp[tid] + rsp[MRB + tid]) + (rsp[2 * MRB + tid] + rsp[3 * MRB + tid]));
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
    if constexpr (GRAM) {
      __syncthreads();
      gram_update<W>(Kt, isr, l15, lq, accB);
      const double* kq = Kt + rg * 32 * MLD + col;
      const double* yq = yisr + rg * 32;
#pragma unroll 8
      for (int i = 0; i < 32; ++i) csum += kq[i * MLD] * yq[i];
    }
  }
  double* part = a.part + (int64_t)blockIdx.x * (GRAM ? M1LEN : W1LEN);
  double* ptail = part;
  if constexpr (GRAM) {
    store_gram<W>(part, accB, l15, lq);
    __syncthreads();
    Kt[rg * MLD + col] = csum;
    __syncthreads();
    if (tid < MP) part[MGLEN + tid] = Kt[tid] + Kt[MLD + tid];
    ptail = part + MGLEN + MP;
  }
  if (W == 0) {
    p_log = sum64(p_log);
    p_y2 = sum64(p_y2);
    p_isr = sum64(p_isr);
    if (lane == 0) {
      ptail[0] = p_log;
      ptail[1] = p_y2;
      ptail[2] = p_isr;
      ptail[3] = 0.0;
    }
  }
}

template <int MPV, int DT>
__global__ __launch_bounds__(256) void mid_pass1_kernel(MidPass1Args a) {
  extern __shared__ __attribute__((aligned(16))) double mid_lds[];
  switch (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) {  // (a scalar: the branch is uniform, no lane masking)
    case 0: mid_pass1_body<MPV, DT, 0>(a, mid_lds); break;
    case 1: mid_pass1_body<MPV, DT, 1>(a, mid_lds); break;
    case 2: mid_pass1_body<MPV, DT, 2>(a, mid_lds); break;
    default: mid_pass1_body<MPV, DT, 3>(a, mid_lds); break;
  }
}

// the 128 x 128 tile of an exchange buffer from the workgroups' upper 16 x 16 tiles (zero below the diagonal 16-blocks:
// every consumer reads upper triangles), summed in workgroup order
__device__ __forceinline__ double mid_tile_entry(const double* __restrict__ part, int64_t plen, int ng, int r, int c) {
  const int it = r >> 4, jt = c >> 4;
  if (it > jt) return 0.0;
  return sum_parts(part + tix(it, jt) * 256 + (r & 15) * 16 + (c & 15), plen, ng);
}

__global__ __launch_bounds__(256) void mid_reduce1_kernel(const double* __restrict__ part, int ng, double* __restrict__ tile,
                                                          double* __restrict__ cvec, double* __restrict__ tail) {
  const int tid = threadIdx.x;
  if (blockIdx.x < MP * MP / 256) {
    const int idx = blockIdx.x * 256 + tid;
    tile[idx] = mid_tile_entry(part, M1LEN, ng, idx / MP, idx % MP);
    return;
  }
  if (tid < MP) cvec[tid] = sum_parts(part + MGLEN + tid, M1LEN, ng);
  else if (tid >= 192 && tid < 196) tail[tid - 192] = sum_parts(part + MGLEN + MP + (tid - 192), M1LEN, ng);
}

// ---------------------------------------------------------------------------------------------------------------- pass 2
// NMT: 16-row tiles of the moment matrix [1 | p_1 .. p_d | x_big,1 .. x_big,D] (1 + d + D <= 16 NMT, NMT <= 2)
template <int MPV, int DT, int NMT, int W>
__device__ __forceinline__ void mid_pass2_body(const MidPass2Args& a, double* lds) {
  using G = MidGeo<MPV>;
  constexpr int MP = G::MP, MLD = G::MLD, MRB = G::MRB, NJ = G::NJ, RT = G::RT;
  constexpr bool GRAM = G::GRAM;
  constexpr int LDM = 16 * NMT + 2;
  double* const Vt = lds;               // [MRB][MLD] V of the block
  double* const Qt = Vt + MRB * MLD;    // [MRB][MLD] Q', then X~, then X, then E, each in place
  double* const Mx = Qt + MRB * MLD;    // [MRB][LDM] moment matrix of the block's rows: column 0 ones, 1 .. d the points
  double* const isr = Mx + MRB * LDM;   // [MRB] per-row values of the block
  double* const vr = isr + MRB;
  double* const wr = vr + MRB;
  double* const q2p = wr + MRB;         // [4][MRB] row sums of Q'.^2 / Q' b over each wavefront's columns
  double* const qbp = q2p + 4 * MRB;    // [4][MRB]
  double* const bv = qbp + 4 * MRB;     // [MP] b
  double* const tt = bv + MP;           // [MP] t~
  double* const red = tt + MP;          // [16] scratch of the final scalar reductions
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lq = lane >> 4;
  const ExpK ek = exp_consts();
  const int d = a.d, D = a.D, nmom = 1 + d + D;
  for (int idx = tid; idx < MP; idx += 256) {
    bv[idx] = a.bvec[idx];
    tt[idx] = a.ttil[idx];
  }
  const int col = tid % MP, rg = tid / MP;
  const bool live_c = col < a.m;
  // Points and inducing coordinates enter this pass shifted by the centroid of the inducing points (a.shift): the moments
  // sum_r p_kr E_rc of data far from the origin (offset 1e4 at unit spread) would otherwise carry the offset through 3000
  // additions of mixed sign and lose it again in the host's p - z; mid_reduce2_kernel adds shift_k sum_r E_rc back once,
  // after the workgroups' sums (as grad_mfma_kernel does per slab).  The distances are differences either way.
  double z[DT];
#pragma unroll
  for (int k = 0; k < DT; ++k) z[k] = (k < d && live_c) ? a.Z[(int64_t)col * d + k] - a.shift[k] : 0.0;
  sd4 accG[GRAM ? 9 : 1];
#pragma unroll
  for (int q = 0; q < (GRAM ? 9 : 1); ++q) accG[q] = sd4{0.0, 0.0, 0.0, 0.0};
  sd4 accM[NMT][NJ];  // moments of E: rows 16 mt .., column tiles W + 4 jj
#pragma unroll
  for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) accM[mt][jj] = sd4{0.0, 0.0, 0.0, 0.0};
  double sE = 0.0, sED = 0.0;
  double p_v = 0.0, p_is = 0.0, p_res = 0.0, p_v1 = 0.0;
  const int nblk = a.rows_p / MRB;
  for (int b = blockIdx.x; b < nblk; b += gridDim.x) {
    const int r0 = b * MRB;
    __syncthreads();
    for (int idx = tid; idx < MRB * 16 * NMT; idx += 256) {
      const int r = idx / (16 * NMT), q = idx % (16 * NMT);
      double v = 0.0;
      if (r0 + r < a.rows) {
        if (q == 0) v = 1.0;
        else if (q <= d) v = a.pts[(int64_t)(r0 + r) * d + (q - 1)] - a.shift[q - 1];
        else if (q < nmom) v = a.big[(int64_t)(r0 + r) * D + (q - 1 - d)];
      }
      Mx[r * LDM + q] = v;
    }
    {  // V of the block (8192 entries): sixteen 16-byte loads per thread, all issued before the first store
      double2 v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int idx = tid + 256 * j, r = idx / (MP / 2), c2 = (idx % (MP / 2)) * 2;
        v[j] = *reinterpret_cast<const double2*>(a.V + (int64_t)(r0 + r) * MP + c2);
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int idx = tid + 256 * j, r = idx / (MP / 2), c2 = (idx % (MP / 2)) * 2;
        *reinterpret_cast<double2*>(Vt + r * MLD + c2) = v[j];
      }
    }
    if (tid < MRB) isr[tid] = a.is[r0 + tid];
    const bool rowlive = tid < MRB && r0 + tid < a.rows;  // the row phase below: one thread per row
    const double rreg = rowlive ? a.r[r0 + tid] : 0.0;
    const double yreg = (rowlive && a.y) ? a.y[r0 + tid] : 0.0;
    __syncthreads();
    sd4 acc[RT][NJ];
    gmem_ptr rinv = (gmem_ptr)a.rinv, rinvT = (gmem_ptr)a.rinvT, uinvT = (gmem_ptr)a.uinvT;  // (laundered per block, see pass 1)
    asm volatile("" : "+s"(rinv), "+s"(rinvT), "+s"(uinvT));
    tri_nn<MPV, W>(Vt, rinv, l15, lq, acc);  // Q' = V R~^-1
    double* const qq = Qt + lq * MLD + l15;        // element r of acc[i][jj] is row 16 i + lq + 4 r, column 16 J_jj + l15
    const double* const vq = Vt + lq * MLD + l15;
    {
      double bj[NJ];
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) bj[jj] = bv[16 * wave_tile<MPV, W>(jj) + l15];
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          double s2 = 0.0, sb = 0.0;
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj) {
            qq[(16 * i + 4 * r) * MLD + 16 * wave_tile<MPV, W>(jj)] = acc[i][jj][r];
            s2 += acc[i][jj][r] * acc[i][jj][r];
            sb += acc[i][jj][r] * bj[jj];
          }
          s2 = sum16(s2);
          sb = sum16(sb);
          if (l15 == 0) {
            (q2p + lq)[W * MRB + 16 * i + 4 * r] = s2;
            (qbp + lq)[W * MRB + 16 * i + 4 * r] = sb;
          }
        }
    }
    __syncthreads();
    if (tid < MRB) {  // q_diag, w, v (as pass2_rows_kernel)
      const int row = r0 + tid;
      double w = 0.0, v = 0.0, es = 0.0;
      if (row < a.rows) {
        const double is = isr[tid], rr = rreg;
        const double q2 = (q2p[tid] + q2p[MRB + tid]) + (q2p[2 * MRB + tid] + q2p[3 * MRB + tid]);
        const double sb = (qbp[tid] + qbp[MRB + tid]) + (qbp[2 * MRB + tid] + qbp[3 * MRB + tid]);
        const double qd = is * q2;
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
    }
    tri_nt<MPV, W>(Qt, rinvT, l15, lq, acc);  // Q' R~^-T
    __syncthreads();  // every wavefront is done reading Q'; the row values are there
    {
      double tj[NJ];
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) tj[jj] = tt[16 * wave_tile<MPV, W>(jj) + l15];
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {  // X~ = diag(is) Q' R~^-T - diag(v) V - w t~^T
          const double is = (isr + lq)[16 * i + 4 * r], v = (vr + lq)[16 * i + 4 * r], w = (wr + lq)[16 * i + 4 * r];
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj) {
            const int off = (16 * i + 4 * r) * MLD + 16 * wave_tile<MPV, W>(jj);
            qq[off] = is * acc[i][jj][r] - v * vq[off] - w * tj[jj];
          }
        }
    }
    __syncthreads();
    tri_nt<MPV, W>(Qt, uinvT, l15, lq, acc);  // X = X~ U^-T
    __syncthreads();
    {
      double* xg = a.X ? a.X + (int64_t)(r0 + lq) * MP + l15 : nullptr;
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj) {
            qq[(16 * i + 4 * r) * MLD + 16 * wave_tile<MPV, W>(jj)] = acc[i][jj][r];
            if (xg) xg[(16 * i + 4 * r) * MP + 16 * wave_tile<MPV, W>(jj)] = acc[i][jj][r];
          }
    }
    if constexpr (GRAM) gram_update<W>(Vt, vr, l15, lq, accG);  // G~ part = V^T diag(v) V
    __syncthreads();
    // E = X .* K of the block (K recomputed from the staged points), in place; sum E, sum E |x - z|^2
    {
      const double* mq = Mx + rg * 32 * LDM + 1;
      double* eq = Qt + rg * 32 * MLD + col;
      const int rlive = a.rows - r0 - rg * 32;
#pragma unroll 4
      for (int i = 0; i < 32; ++i) {
        double dist = 0.0;
#pragma unroll
        for (int k = 0; k < DT; ++k) {
          const double diff = (k < d) ? mq[i * LDM + k] - z[k] : 0.0;  // (uniform test; the points are columns 1 .. d)
          dist = dist + diff * diff;
        }
        const double e = (live_c && i < rlive) ? eq[i * MLD] * exp_fast(a.cp.log_sf2 + a.cp.inv_ell2_05 * dist, ek) : 0.0;
        eq[i * MLD] = e;
        sE += e;
        sED += e * dist;
      }
    }
    __syncthreads();
    // moments of E against [1 | p | x_big]: accM[mt][jj] += Mx^T E, tile rows 16 mt .., column tiles W + 4 jj
    const double* const mfq = Mx + lq * LDM + l15;
#pragma unroll
    for (int h = 0; h < MRB / 16; ++h) {
      double af[4][NMT], bf[4][NJ];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int k4 = 4 * (4 * h + s);
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) af[s][mt] = mfq[k4 * LDM + 16 * mt];
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) bf[s][jj] = qq[k4 * MLD + 16 * (W + 4 * jj)];
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj) accM[mt][jj] = mfma_f64(af[s][mt], bf[s][jj], accM[mt][jj]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const int plen = GRAM ? m2len(d, D) : w2len(d, D);
  double* part = a.part + (int64_t)blockIdx.x * plen;
  double* pcol = part;
  if constexpr (GRAM) {
    store_gram<W>(part, accG, l15, lq);
    pcol = part + MGLEN;
  }
#pragma unroll
  for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int q = 16 * mt + lq + 4 * r;
      if (q < nmom) {
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) pcol[q * MP + 16 * (W + 4 * jj) + l15] = accM[mt][jj][r];
      }
    }
  double* ptail = pcol + nmom * MP;
  sE = sum64(sE);
  sED = sum64(sED);
  __syncthreads();
  if (lane == 0) {
    red[W] = sE;
    red[4 + W] = sED;
  }
  __syncthreads();
  if (W == 0) {
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

template <int MPV, int DT, int NMT>
__global__ __launch_bounds__(256) void mid_pass2_kernel(MidPass2Args a) {
  extern __shared__ __attribute__((aligned(16))) double mid_lds[];
  switch (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) {
    case 0: mid_pass2_body<MPV, DT, NMT, 0>(a, mid_lds); break;
    case 1: mid_pass2_body<MPV, DT, NMT, 1>(a, mid_lds); break;
    case 2: mid_pass2_body<MPV, DT, NMT, 2>(a, mid_lds); break;
    default: mid_pass2_body<MPV, DT, NMT, 3>(a, mid_lds); break;
  }
}

// exchange-2 buffer from the pass-2 partials.  One tile (tile != null): every entry written -- the 128 x 128 tile, the column
// block (col_rows x mp; rows 0 .. d + D carry sums, the rest -- the multiscale rows of a Cov_se_fat layout -- zero), the `Proj
// second term (zero: proj_term2_kernel adds it afterwards) and the scalar tail.  Two tiles (tile == null: the partials
// carry no Gram part, G~ comes from the engine's launch): everything but the tiles.
__global__ __launch_bounds__(256) void mid_reduce2_kernel(const double* __restrict__ part, int ng, int mp, int d, int D,
                                                          int col_rows, int nproj, const double* __restrict__ shift,
                                                          double* __restrict__ tile, double* __restrict__ colblk,
                                                          double* __restrict__ proj, double* __restrict__ tail) {
  const int nmom = 1 + d + D;
  const int plen = tile ? m2len(d, D) : w2len(d, D), pbase = tile ? MGLEN : 0;
  int idx = blockIdx.x * 256 + threadIdx.x;
  const int ntile = tile ? MP * MP : 0, ncol = col_rows * mp;
  if (idx < ntile) {
    tile[idx] = mid_tile_entry(part, plen, ng, idx / MP, idx % MP);
  } else if ((idx -= ntile) < ncol) {
    const int q = idx / mp, c = idx % mp;
    double val = q < nmom ? sum_parts(part + pbase + q * mp + c, plen, ng) : 0.0;
    // rows 1 .. d: the moments were taken against p - shift (mid_pass2_body); sum_r p_kr E_rc = that + shift_k sum_r E_rc
    if (q >= 1 && q <= d) val += shift[q - 1] * sum_parts(part + pbase + c, plen, ng);
    colblk[idx] = val;
  } else if ((idx -= ncol) < nproj) {
    proj[idx] = 0.0;
  } else if ((idx -= nproj) < 8) {
    tail[idx] = sum_parts(part + pbase + nmom * mp + idx, plen, ng);
  }
}

// ---------------------------------------------------------------------------------------------------------------- finish
// The m x m work of do_finish_enqueue for one or two 128-blocks (as small_finish_kernel for 64 x 64 corners):
//   B~^-1 = R~^-1 R~^-T (Utils.ichol, lib/utils.ml:110-113),  W~ = I - B~^-1 - t~ t~^T - G~,  W = U^-1 W~ U^-T
//   (lib/fitc_gp.ml:1196-1203), the trace terms of W against K_m and its derivatives (km_traces_kernel: :956-973,
//   lib/utils.ml:196-220), diag W, and the tails of both exchange buffers gathered behind the result block.
// Two launches of one workgroup per 16-row block -- a single workgroup doing all three 128^3 products takes 70 us (18 us of
// MFMAs on one CU and a memory round trip per k-batch), the eight together a few:
//   mid_finish1: rows I of B~^-1 -> rows I of W~ (LDS) -> rows I of Y = W~ U^-T (memory)
//   mid_finish2: rows I of W = U^-1 Y, and -- W and K_m are symmetric -- the trace terms of COLUMNS I from those rows,
//                so no sum across workgroups is left
// Every fragment that comes from memory is requested a chunk of four k-batches at a time, before the chunk's first MFMA
// (nothing here sits in a loop that could hoist them).  G~ is read through gprhip_exchange_offset's packed-tile layout.
template <int MPV>
__device__ __forceinline__ int64_t mid_packed_off(int r, int c) {  // r <= c
  return MPV == 128 ? (int64_t)r * 128 + c : packed_upper_off(r, c);
}

// acc[jj] = rows of Arow (LDS [16][MLD]) times B^T restricted to k >= 16 max(kmin_row, j), given Bt = B^T in memory:
// out[i][c] = sum_k A[i][k] Bt[k][c]
template <int MPV>
__device__ __forceinline__ void mid_rows_times_tri_t(const double* Arow, const double* __restrict__ Bt, int kmin_row, int wv,
                                                      int l15, int lq, sd4 (&acc)[MidGeo<MPV>::NJ]) {
  using G = MidGeo<MPV>;
  constexpr int NJ = G::NJ;
  int jt[NJ];
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj) {
    jt[jj] = wave_tile_rt(MPV, wv, jj);
    acc[jj] = sd4{0.0, 0.0, 0.0, 0.0};
  }
  const double* aq = Arow + l15 * G::MLD + lq;
#pragma unroll 1
  for (int ch = kmin_row / 4; ch < G::MCT / 4; ++ch) {
    double bf[4][4][NJ];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) {
        // (unconditional: a load under a uniform branch ends in a wait at the branch's end -- sixteen round trips per chunk
        //  instead of one, 70 of this kernel's 80 us at m = 256; what a tile does not need yet is fetched and left unused)
        const int kb = 4 * ch + kk;
        const double* bq = Bt + (int64_t)(16 * kb + lq) * G::MP + 16 * jt[jj] + l15;
#pragma unroll
        for (int s = 0; s < 4; ++s) bf[kk][s][jj] = bq[4 * s * G::MP];
      }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int kb = 4 * ch + kk;
      if (kb < kmin_row) continue;  // (uniform)
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj)
        if (kb >= jt[jj]) {  // (uniform)
#pragma unroll
          for (int s = 0; s < 4; ++s) acc[jj] = mfma_f64(aq[16 * kb + 4 * s], bf[kk][s][jj], acc[jj]);
        }
    }
  }
}

template <int MPV>
__global__ __launch_bounds__(256) void mid_finish1_kernel(MidFinishArgs a) {
  using G = MidGeo<MPV>;
  constexpr int MP = G::MP, MLD = G::MLD, NJ = G::NJ;
  __shared__ __attribute__((aligned(16))) double Rrow[16 * MLD];  // rows I of R~^-1, then of W~
  __shared__ double tt[MP];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, lq = lane >> 4;
  const int I = blockIdx.x;
  for (int idx = tid; idx < MP; idx += 256) tt[idx] = a.ttil[idx];
  for (int idx = tid; idx < 16 * (MP / 2); idx += 256) {
    const int r = idx / (MP / 2), c2 = (idx % (MP / 2)) * 2;
    *reinterpret_cast<double2*>(Rrow + r * MLD + c2) = *reinterpret_cast<const double2*>(a.rinv + (int64_t)(16 * I + r) * MP + c2);
  }
  if (I == 0)
    for (int64_t i = tid; i < a.n_gather; i += 256) a.ex[i] = a.gather_from[i];
  __syncthreads();
  sd4 acc[NJ];
  mid_rows_times_tri_t<MPV>(Rrow, a.rinvT, I, wv, l15, lq, acc);  // B~^-1[i][j] = sum_{k >= max(i, j)} Ri[i][k] Ri[j][k]
  __syncthreads();  // every wavefront is done reading the rows of R~^-1
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int li = lq + 4 * r, row = 16 * I + li, c = 16 * wave_tile_rt(MPV, wv, jj) + l15;
      const int rr = min(row, c), cc = max(row, c);  // G~ is valid in the upper triangle: mirrored, as build_w_kernel
      Rrow[li * MLD + c] = (row == c ? 1.0 : 0.0) - acc[jj][r] - tt[row] * tt[c] - a.g[mid_packed_off<MPV>(rr, cc)];
    }
  __syncthreads();
  mid_rows_times_tri_t<MPV>(Rrow, a.uinvT, 0, wv, l15, lq, acc);  // Y[i][j] = sum_{k >= j} W~[i][k] Ui[j][k]
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      a.ybuf[(int64_t)(16 * I + lq + 4 * r) * MP + 16 * wave_tile_rt(MPV, wv, jj) + l15] = acc[jj][r];
}

template <int MPV, int DT>
__global__ __launch_bounds__(256) void mid_finish2_kernel(MidFinishArgs a) {
  using G = MidGeo<MPV>;
  constexpr int MP = G::MP, MLD = G::MLD, NJ = G::NJ, NC = MP / 16;  // NC: columns per thread of the trace phase
  extern __shared__ __attribute__((aligned(16))) double mid_lds[];
  double* const Urow = mid_lds;          // [16][MLD] rows I of U^-1, then of W
  double* const zs = Urow + 16 * MLD;    // [MP][DT]
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, lq = lane >> 4;
  const int I = blockIdx.x, d = a.d, m = a.m;
  for (int idx = tid; idx < 16 * (MP / 2); idx += 256) {
    const int r = idx / (MP / 2), c2 = (idx % (MP / 2)) * 2;
    *reinterpret_cast<double2*>(Urow + r * MLD + c2) = *reinterpret_cast<const double2*>(a.uinv + (int64_t)(16 * I + r) * MP + c2);
  }
  for (int idx = tid; idx < MP * DT; idx += 256) {
    const int c = idx / DT, k = idx % DT;
    zs[idx] = (k < d && c < m) ? a.Z[(int64_t)c * d + k] : 0.0;
  }
  // K_m entries of the trace phase (thread = (row i of the block, NC columns)): requested now, used at the end
  const int ti = tid >> 4, tp = tid & 15, crow = 16 * I + ti;
  double kreg[NC];
#pragma unroll
  for (int j = 0; j < NC; ++j) kreg[j] = (crow < m && NC * tp + j < m) ? a.km[(int64_t)crow * MP + NC * tp + j] : 0.0;
  __syncthreads();
  // W[i][j] = sum_{k >= i} Ui[i][k] Y[k][j]: the batches kb >= I; column tiles wv + 4 jj
  sd4 acc[NJ];
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj) acc[jj] = sd4{0.0, 0.0, 0.0, 0.0};
  {
    const double* aq = Urow + l15 * MLD + lq;
    const double* yq = a.ybuf + (int64_t)lq * MP + l15 + 16 * wv;
#pragma unroll 1
    for (int ch = I / 4; ch < G::MCT / 4; ++ch) {
      double bf[4][4][NJ];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int kb = 4 * ch + kk;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj) bf[kk][s][jj] = yq[(int64_t)(16 * kb + 4 * s) * MP + 64 * jj];  // (unconditional, as above)
      }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int kb = 4 * ch + kk;
        if (kb < I) continue;  // (uniform)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const double af = aq[16 * kb + 4 * s];
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj) acc[jj] = mfma_f64(af, bf[kk][s][jj], acc[jj]);
        }
      }
    }
  }
  __syncthreads();  // every wavefront is done reading the rows of U^-1
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int li = lq + 4 * r, c = 16 * (wv + 4 * jj) + l15;
      Urow[li * MLD + c] = acc[jj][r];
      a.wmat[(int64_t)(16 * I + li) * MP + c] = acc[jj][r];
    }
  __syncthreads();
  // trace terms of column crow (= row crow: W and K_m are symmetric): sum over r of W[crow][r] K_m[crow][r] f(z_r - z_crow)
  double g[DT], s0 = 0.0, s1 = 0.0;
#pragma unroll
  for (int k = 0; k < DT; ++k) g[k] = 0.0;
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    const int r = NC * tp + j;
    const double wk = Urow[ti * MLD + r] * kreg[j];  // (0 beyond the real rows and columns)
    s0 += wk;
    double dist = 0.0;
#pragma unroll
    for (int k = 0; k < DT; ++k) {
      const double df = zs[r * DT + k] - zs[crow * DT + k];
      dist += df * df;
      g[k] += wk * df;
    }
    s1 += wk * dist;
  }
  s0 = sum16(s0);
  s1 = sum16(s1);
#pragma unroll
  for (int k = 0; k < DT; ++k) g[k] = sum16(g[k]);
  if (tp == 0) {
    a.kmred[crow] = s0;
    a.kmred[MP + crow] = s1;
#pragma unroll
    for (int k = 0; k < DT; ++k)
      if (k < d) a.kmred[(int64_t)(2 + k) * MP + crow] = g[k];
    for (int q = 2 + d; q < a.km_rows; ++q) a.kmred[(int64_t)q * MP + crow] = 0.0;
    if (a.wdiag) a.wdiag[crow] = Urow[ti * MLD + crow];
  }
  // The result block goes to the host from HERE: the workgroup that finishes last copies it into the pinned mirror (which the
  // device addresses directly), instead of one or two copy launches behind the kernel -- 4 us at one tile, and at two tiles
  // (43 KB) a 17 us gap in front of them besides (profiles/r06_timeline_n2560_m256.txt).  Writers fence at agent scope before
  // they count themselves in (the XCDs' L2s are not coherent with each other), the last one fences again before it reads.
  if (a.res_host) {
    __shared__ int is_last;
    __threadfence();
    __syncthreads();
    if (tid == 0) is_last = atomicAdd(a.done_ctr, 1) == (int)gridDim.x - 1;
    __syncthreads();
    if (is_last) {
      __threadfence();
      for (int64_t i = tid; i < a.res_total; i += 256) a.res_host[i] = __builtin_nontemporal_load(a.res_dev + i);
      __syncthreads();  // (a1_host lies inside the block just copied: behind it)
      if (a.a1_tail && tid < 4) a.a1_host[tid] = a.a1_tail[tid];
      if (tid == 0) *a.done_ctr = 0;  // (for the next evaluation; stream order separates the launches)
    }
  }
}

// U^-T and R~^-T beside U^-1 and R~^-1 (one launch, before pass 2 of a gradient evaluation): see tri_nt
__global__ __launch_bounds__(256) void mid_transpose2_kernel(const double* __restrict__ u, const double* __restrict__ r, int mp,
                                                             double* __restrict__ ut, double* __restrict__ rt) {
  __shared__ double t[32][33];
  const double* src = blockIdx.z == 0 ? u : r;
  double* dst = blockIdx.z == 0 ? ut : rt;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5, r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
#pragma unroll
  for (int i = 0; i < 4; ++i) t[ty + 8 * i][tx] = src[(int64_t)(r0 + ty + 8 * i) * mp + c0 + tx];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) dst[(int64_t)(c0 + ty + 8 * i) * mp + r0 + tx] = t[tx][ty + 8 * i];
}

void launch_mid_transposes(const double* uinv, const double* rinv, int mp, double* uinvT, double* rinvT, hipStream_t s) {
  hipLaunchKernelGGL(mid_transpose2_kernel, dim3(mp / 32, mp / 32, 2), dim3(256), 0, s, uinv, rinv, mp, uinvT, rinvT);
  GPR_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------------- host
static size_t mid_lds1(int mp, int DT) {
  const int rb = mp == 128 ? 64 : 32;
  return (size_t)(rb * (mp + 2) + rb * DT + 2 * rb + 4 * rb) * sizeof(double);
}
static size_t mid_lds2(int mp, int NMT) {
  const int rb = mp == 128 ? 64 : 32;
  return (size_t)(2 * rb * (mp + 2) + rb * (16 * NMT + 2) + 3 * rb + 8 * rb + 2 * mp + 16) * sizeof(double);
}
static size_t mid_lds3(int mp, int DT) { return (size_t)(16 * (mp + 2) + mp * DT) * sizeof(double); }

template <typename F>
static void mid_dispatch(int d, F&& go) {
  if (d <= 4) go(std::integral_constant<int, 4>{});
  else if (d <= 8) go(std::integral_constant<int, 8>{});
  else go(std::integral_constant<int, 16>{});
}

// ---------------------------------------------------------------------------------------- Gram accumulation, two tiles
// G = V^T diag(w) V on the three upper 128-tiles (0,0), (0,1), (1,1) and c = V^T y, V the shard's resident [rows][256].
// The engine's SYRK-shaped launch does this for any size; on the few thousand rows of the reference's default regime
// (n = 1290 .. 2560 for m = 129 .. 256) its fixed cost -- workgroup entry, LDS staging, a 128-deep first stage: 28 us -- is
// the whole launch, and its slice sum and column-sum reduction are two more launches.  Here: workgroup (tile, row slice),
// eight wavefronts = two halves of the slice's rows x the four 64 x 64 quadrants of the tile, 16 accumulator tiles each; both
// operands of an MFMA are rows of V (lane (l15, lq) of k-step s reads V[k + lq][column + l15]: four 128-byte runs per load),
// fetched straight from memory four k-steps ahead of their use; the halves meet through LDS and the slice's partial tile
// goes to memory in the packed layout of the exchange buffers.  mid_gram_reduce_kernel sums the slices in order, 16 loads
// in flight.  Strictly-lower 16 x 16 sub-tiles of the two diagonal tiles are written as zeros, as the engine leaves them.
// Every wavefront fetching its own fragments is four times the L2 traffic of a staged tile, and the compiler joins the
// waits of a round of four k-steps at its top (0.75 us per k-step once the slices are long): good for short slices only --
// the caller uses this pair up to MID_GRAM_ROWS rows and the engine above (profiles/r06_latency_gram.txt).  (An LDS-staged
// version, three chunks of 16 rows in flight, was built and was slower still: whatever is computed from a fetched register
// the scheduler moves up next to its load, across the barriers, and the wait with it.)
constexpr int GRAM_MAX_SLICES = 64;
constexpr int GRAM_TLEN = 3 * 128 * 128;       // the three tiles of one slice
constexpr int GRAM_PLEN = GRAM_TLEN + 256;     // ... and its column sums
constexpr int GRAM_LD = 130;                   // LDS tile of the second half's accumulators
constexpr int GRAM_LDS = 128 * GRAM_LD * 8 + 4 * 64 * 8;
constexpr int GRAM_NB = 4;                     // k-steps of fragments in flight

__global__ __launch_bounds__(512) void mid_gram_kernel(MidGramArgs a, int rps) {
  extern __shared__ __attribute__((aligned(16))) double glds[];
  double* const T = glds;                      // [128][GRAM_LD]
  double* const cs2 = T + 128 * GRAM_LD;       // [2][64] column sums of the second half
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const int t = blockIdx.x, sl = blockIdx.y;
  const int I0 = t == 2 ? 128 : 0, J0 = t == 0 ? 0 : 128;
  const int g = wid >> 2, wr = (wid >> 1) & 1, wc = wid & 1;
  const int hl = rps / 2;                                    // rows per half (a multiple of 4)
  const int kbeg = sl * rps + g * hl, kend = min(a.rows, kbeg + hl);
  const bool sums = a.y != nullptr && t != 1 && wc == 0;     // column sums: diagonal tiles, one wave column
  const double* const ysrc = a.y ? a.y : a.w;
  sd4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = sd4{0.0, 0.0, 0.0, 0.0};
  double cs[4] = {0.0, 0.0, 0.0, 0.0};
  const double* const Va = a.V + I0 + wr * 64 + l15;
  const double* const Vb = a.V + J0 + wc * 64 + l15;
  double fa[GRAM_NB][4], fb[GRAM_NB][4], fw[GRAM_NB], fy[GRAM_NB];
  auto fetch = [&](int slot, int step) {
    const int kr = kbeg + 4 * step + lq;
    const int kw = min(kr, a.rows - 1);   // (rows beyond the half weigh zero; they are read from a real row all the same,
    const int64_t row = (int64_t)kw * 256;  //  so that the zero multiplies something finite whatever the padding holds)
    const bool in = kr < kend;
    const double wv = a.w[kw], yv = ysrc[kw];  // unconditional loads: a load under a branch ends in a wait at its end
    fw[slot] = in ? wv : 0.0;
    fy[slot] = (in && sums) ? yv : 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[slot][i] = Va[row + 16 * i];
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[slot][j] = Vb[row + 16 * j];
  };
  auto work = [&](int slot) {
    double bw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bw[j] = fb[slot][j] * fw[slot];
    if (sums) {
#pragma unroll
      for (int i = 0; i < 4; ++i) cs[i] = fma(fa[slot][i], fy[slot], cs[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = mfma_f64(fa[slot][i], bw[j], acc[i][j]);
  };
  const int nsteps = (max(kend - kbeg, 0) + 3) / 4;
  const int nloop = (nsteps + GRAM_NB - 1) / GRAM_NB * GRAM_NB;   // (whole rounds of the buffers: the extra steps weigh zero)
#pragma unroll
  for (int u = 0; u < GRAM_NB - 1; ++u) fetch(u, u);
#pragma unroll 1
  for (int s0 = 0; s0 < nloop; s0 += GRAM_NB) {
#pragma unroll
    for (int u = 0; u < GRAM_NB; ++u) {
      fetch((u + GRAM_NB - 1) % GRAM_NB, s0 + u + GRAM_NB - 1);
      work(u);
    }
  }
  // the second half's accumulators through LDS, the first half adds its own and stores the slice's partial tile
  if (g == 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) T[(wr * 64 + 16 * i + lq + 4 * r) * GRAM_LD + wc * 64 + 16 * j + l15] = acc[i][j][r];
    if (sums) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        double v = cs[i];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (lq == 0) cs2[wr * 64 + 16 * i + l15] = v;
      }
    }
  }
  __syncthreads();
  if (g == 0) {
    double* const P = a.part + (int64_t)sl * GRAM_PLEN;
    double* const Pt = P + (int64_t)t * 128 * 128;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool keep = t == 1 || wr * 4 + i <= wc * 4 + j;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rr = wr * 64 + 16 * i + lq + 4 * r, cc = wc * 64 + 16 * j + l15;
          Pt[rr * 128 + cc] = keep ? acc[i][j][r] + T[rr * GRAM_LD + cc] : 0.0;
        }
      }
    if (sums) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        double v = cs[i];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (lq == 0) P[GRAM_TLEN + I0 + wr * 64 + 16 * i + l15] = v + cs2[wr * 64 + 16 * i + l15];
      }
    }
  }
}

// tiles[e] = sum over the slices, in order, of their partial e (e < 3 * 128 * 128: the packed upper tiles), cvec likewise
// from the partials' last 256 entries (cvec == null: not wanted); two entries per thread, 16 slices' loads in flight
__global__ __launch_bounds__(256) void mid_gram_reduce_kernel(const double* __restrict__ part, int nslice,
                                                              double* __restrict__ tiles, double* __restrict__ cvec) {
  const int e = (blockIdx.x * 256 + threadIdx.x) * 2;
  if (e >= GRAM_PLEN || (e >= GRAM_TLEN && !cvec)) return;
  const double* p = part + e;
  double s0 = 0.0, s1 = 0.0;
  for (int z = 0; z < nslice; z += 16) {
    double2 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = *reinterpret_cast<const double2*>(p + (int64_t)min(z + u, nslice - 1) * GRAM_PLEN);
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (z + u < nslice) {
        s0 += v[u].x;
        s1 += v[u].y;
      }
    }
  }
  double* out = e < GRAM_TLEN ? tiles + e : cvec + (e - GRAM_TLEN);
  out[0] = s0;
  out[1] = s1;
}

int64_t mid_gram_part_len() { return (int64_t)GRAM_MAX_SLICES * GRAM_PLEN; }

static void mid_attrs() {
  static uint64_t done = 0;
  once_per_device(done, [] {
    auto set = [](const void* f, size_t bytes) {
      GPR_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    };
#define GPRHIP_MID_SET(MPV, DT)                                                                    \
  set(reinterpret_cast<const void*>(&mid_pass1_kernel<MPV, DT>), mid_lds1(MPV, DT));               \
  set(reinterpret_cast<const void*>(&mid_pass2_kernel<MPV, DT, 1>), mid_lds2(MPV, 1));             \
  set(reinterpret_cast<const void*>(&mid_pass2_kernel<MPV, DT, 2>), mid_lds2(MPV, 2));             \
  set(reinterpret_cast<const void*>(&mid_finish2_kernel<MPV, DT>), mid_lds3(MPV, DT));
    GPRHIP_MID_SET(128, 4)
    GPRHIP_MID_SET(128, 8)
    GPRHIP_MID_SET(128, 16)
    GPRHIP_MID_SET(256, 4)
    GPRHIP_MID_SET(256, 8)
    GPRHIP_MID_SET(256, 16)
#undef GPRHIP_MID_SET
    set(reinterpret_cast<const void*>(&mid_gram_kernel), GRAM_LDS);
  });
}

bool mid_path_fits(int m, int mp, int d, int D, int64_t rows, bool ms) {
  // (D: input dimensions in front of a projection, 0 without one; the moment matrix [1 | p | x_big] has at most two 16-row
  //  tiles -- the two row-block tiles of pass 2 leave 30 KB of the LDS for it and the row vectors)
  // Rows of the shard: these kernels are built for latency (one 64- or 32-row batch per workgroup, operands streamed from
  // L2), the engine for throughput.  Measured crossovers (gradient evaluation, d = 8, profiles/r06_latency_mid_vs_engine.txt): one tile
  // -- level with the engine at 10^6 rows (4.84 / 4.77 ms; evidence alone 1.45 / 1.80), ahead below (3 * 10^5: 1.64 /
  // 2.43); two tiles -- ahead to ~4 * 10^4 rows (3 * 10^4: 0.82 / 0.87), behind above (10^5: 1.74 / 1.48, 10^6: 13.6 /
  // 10.7).  Shards of one problem may fall on either side: both families fill the same exchange buffers.
  const int64_t row_limit = (mp == 128) ? (int64_t(1) << 22) : MID_ROWS_TWO_TILES;
  return (mp == 128 || mp == 256) && m <= mp && d <= 16 && 1 + d + D <= 32 && !ms && rows <= row_limit;
}

template <typename F>
static void mid_by_tiles(int mp, F&& go) {
  if (mp == 128) go(std::integral_constant<int, 128>{});
  else go(std::integral_constant<int, 256>{});
}

// One tile: pass 1 and its reduction fill the whole exchange-1 buffer (tile, c~, tail).  Two tiles: the kernel leaves V, r,
// 1/s, y/s and the scalar tail (summed into `tail`); B~ and c~ come from the engine's SYRK-shaped launch over V (caller).
void launch_mid_pass1(const MidPass1Args& a, double* tile, double* cvec, double* tail, hipStream_t s) {
  mid_attrs();
  mid_by_tiles(a.mp, [&](auto mpv) {
    constexpr int MPV = decltype(mpv)::value;
    const int ng = mid_groups<MPV>(a.rows_p);
    mid_dispatch(a.d, [&](auto dt) {
      constexpr int DT = decltype(dt)::value;
      hipLaunchKernelGGL((mid_pass1_kernel<MPV, DT>), dim3(ng), dim3(256), mid_lds1(MPV, DT), s, a);
    });
    if (MPV == 128) hipLaunchKernelGGL(mid_reduce1_kernel, dim3(MP * MP / 256 + 1), dim3(256), 0, s, a.part, ng, tile, cvec, tail);
    else launch_reduce_rows(a.part, ng, W1LEN, tail, 0, s);
  });
  GPR_HIP(hipGetLastError());
}

// One tile: every entry of the exchange-2 buffer is written.  Two tiles (tile == null): everything but the packed tiles,
// which the engine's SYRK-shaped launch over V fills (caller).
void launch_mid_pass2(const MidPass2Args& a, int col_rows, double* tile, double* colblk, double* proj, double* tail,
                      hipStream_t s) {
  mid_attrs();
  const int nmt = (1 + a.d + a.D + 15) / 16;
  const int nproj = a.D * a.d;
  mid_by_tiles(a.mp, [&](auto mpv) {
    constexpr int MPV = decltype(mpv)::value;
    const int ng = mid_groups<MPV>(a.rows_p);
    mid_dispatch(a.d, [&](auto dt) {
      constexpr int DT = decltype(dt)::value;
      if (nmt == 1) hipLaunchKernelGGL((mid_pass2_kernel<MPV, DT, 1>), dim3(ng), dim3(256), mid_lds2(MPV, 1), s, a);
      else hipLaunchKernelGGL((mid_pass2_kernel<MPV, DT, 2>), dim3(ng), dim3(256), mid_lds2(MPV, 2), s, a);
    });
    const int nout = (MPV == 128 ? MP * MP : 0) + col_rows * MPV + nproj + 8;
    hipLaunchKernelGGL(mid_reduce2_kernel, dim3((nout + 255) / 256), dim3(256), 0, s, a.part, ng, MPV, a.d, a.D, col_rows, nproj,
                       a.shift, MPV == 128 ? tile : nullptr, colblk, proj, tail);
  });
  GPR_HIP(hipGetLastError());
}

void launch_mid_finish(const MidFinishArgs& a, hipStream_t s) {
  mid_attrs();
  mid_by_tiles(a.mp, [&](auto mpv) {
    constexpr int MPV = decltype(mpv)::value;
    hipLaunchKernelGGL((mid_finish1_kernel<MPV>), dim3(MPV / 16), dim3(256), 0, s, a);
    mid_dispatch(a.d, [&](auto dt) {
      constexpr int DT = decltype(dt)::value;
      hipLaunchKernelGGL((mid_finish2_kernel<MPV, DT>), dim3(MPV / 16), dim3(256), mid_lds3(MPV, DT), s, a);
    });
  });
  GPR_HIP(hipGetLastError());
}

// tiles: the three packed upper tiles of an exchange buffer; cvec: its 256 column sums (pass 1) or null
void launch_mid_gram(const MidGramArgs& a, double* tiles, double* cvec, hipStream_t s) {
  mid_attrs();
  // slices of about 48 rows (24 per half: six k-steps, i.e. little more than the fragments in flight), at most 64 of them;
  // rows per slice a multiple of 8
  int nslice = std::max(1, std::min(GRAM_MAX_SLICES, a.rows / 48));
  int rps = ((a.rows + nslice - 1) / nslice + 7) / 8 * 8;
  nslice = (a.rows + rps - 1) / rps;
  MidGramArgs b = a;
  if (!cvec) b.y = nullptr;
  hipLaunchKernelGGL(mid_gram_kernel, dim3(3, nslice), dim3(512), GRAM_LDS, s, b, rps);
  hipLaunchKernelGGL(mid_gram_reduce_kernel, dim3((GRAM_PLEN / 2 + 255) / 256), dim3(256), 0, s, a.part, nslice, tiles, cvec);
  GPR_HIP(hipGetLastError());
}

}  // namespace gprhip
