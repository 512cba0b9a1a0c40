// MFMA contraction engine for gfx950: v_mfma_f64_16x16x4_f64 (fp64) and v_mfma_f32_16x16x4_f32 (fp32).
//
// One kernel family covers every n x m x m / m x m x m contraction on the FITC
// path: right-side triangular
// multiplies (the TRSM/TRMM rows), SYRK-shaped accumulations over training points,
// and the trailing updates of the blocked Cholesky / triangular inverse.
//
// All matrices are dense row-major whose dimensions the caller has padded:
// M, N multiples of 128 (TILE), K a multiple of the stage depth (16 for fp64, 32 for fp32).
#pragma once
#include "common.h"

namespace gprhip {

enum GemmOp : int {
  OP_NN = 0,  // C[i][j] = sum_k A[i][k]   * B[k][j]     (A row-major M x K, B row-major K x N)
  OP_NT = 1,  // C[i][j] = sum_k A[i][k]   * B[j][k]     (B row-major N x K)
  OP_TN = 2,  // C[i][j] = sum_k s[k]*A[k][i] * B[k][j]  (A row-major K x M, B row-major K x N)
};

// Which k-tiles a block visits: lets triangular operands skip their zero blocks.
enum GemmTri : int {
  TRI_NONE = 0,
  TRI_KHI_BN = 1,   // k <  end of the block's column tile   (NN, B upper triangular)
  TRI_KLO_BN = 2,   // k >= start of the block's column tile (NT, B upper triangular)
  TRI_KLO_BM = 3,   // k >= start of the block's row tile    (NN, A upper triangular)
  TRI_KLO_MAX = 4,  // k >= max(row tile, column tile) start (NT, A and B upper triangular)
  TRI_BAND = 6,     // row tile start <= k < column tile end      (NN, A and B upper triangular)
  TRI_KHI_MIN = 5,  // k <  min(row tile, column tile) end   (TN, A and B upper triangular)
};

template <typename T>
struct GemmArgsT {
  const T* A = nullptr;
  int64_t lda = 0;
  const T* B = nullptr;
  int64_t ldb = 0;
  T* C = nullptr;
  int64_t ldc = 0;
  int M = 0, N = 0, K = 0;
  double alpha = 1.0;
  double beta = 0.0;              // 0: overwrite, otherwise C = alpha*AB + beta*C
  const T* scale_k = nullptr;     // OP_TN only: per-k weight on the A operand (may be null); length K, 16-byte aligned
  int tri = TRI_NONE;
  int upper_only = 0;    // compute only tiles with row tile <= column tile
  int order = 0;         // block -> tile order of full grids (see tile_of_block)
  int ipw = 1;           // items per workgroup (set by launch_gemm: 2 for the paired order 3)
  // set by launch_gemm with ipw == 2: the first pair_split workgroups run column-tile PAIRS over the first M/128 - tail_panels row
  // panels; the workgroups behind them run the SINGLE tiles of the last tail_panels row panels, longest k-range first -- short
  // items at the end of the launch fill the slots the last round of equal-length pairs would leave idle
  int pair_split = 0, tail_panels = 0;
  int desc2 = 0;         // set by launch_gemm (paired NN products with a triangular B): the second, short item of a pair walks
                         // its k-range downwards, so that the eight short items of a row panel read the same A block at the
                         // same time (GPRHIP_NN_DESC=0 switches it off)
  int syrk = 0;          // set by launch_gemm: weighted TN launch with A == B and upper_only (diagonal tiles skip their lower sub-tiles)
  int sgroup = 16;       // set by launch_gemm: slices per group of the SYRK item order (tile_of_block); 1 = slice by slice
  int dslices = 0;       // set by launch_gemm: k-slices of the diagonal tiles of such a launch (gemm_syrk_diag_slices), 0 = as kslices
  int nbatch = 1;        // independent problems of identical shape: gridDim.y, pointers advance by the strides
  int64_t batch_a = 0, batch_b = 0, batch_c = 0;
  int kslices = 1;       // split K over the grid; slice z writes C + z*slice_stride
  int64_t slice_stride = 0;
  // optional first phase of an OP_NT item (two products into one accumulator tile, one epilogue): with A2 != null the
  // tile is   acc = A2 B2^T;   acc[i][:] *= -mid_num[i] / mid_den[i]  (0 where mid_den[i] == 0);   acc += A B^T
  // over the same triangular k-range; A2 / B2 share the leading dimensions (and alignment) of A / B.
  // (X = diag(is) Q' R^-T - diag(v) V U^-T - w t^T of the gradient pass in one launch: lib/fitc_gp.ml:931-939, :1204-1206)
  const T* A2 = nullptr;
  const T* B2 = nullptr;
  const double* mid_num = nullptr;
  const double* mid_den = nullptr;
  // optional fused epilogue (when epi_rows_a != null): C[i][j] = ra[i]*acc - rb[i]*M[i][j] - rc[i]*cv[j]
  // (epi_mat == null: no M term, C[i][j] = ra[i]*acc - rc[i]*cv[j])
  const double* epi_rows_a = nullptr;
  const double* epi_rows_b = nullptr;
  const double* epi_rows_c = nullptr;
  const double* epi_col = nullptr;
  const T* epi_mat = nullptr;
  int64_t epi_ldm = 0;
  // optional per-row reductions of the result tile, fused into the plain-store epilogue:
  //   rp_sumsq[(P*bn + wc) * M + row] = sum of C[row][c]^2 over the columns wave column wc owns in tile bn,
  //   P = gemm_row_parts_per_tile() wave columns per tile
  //   rp_dot  [same index]            = sum of C[row][c] * rp_vec[c] over the same columns
  // (part-major: the 64 lanes of a wavefront write 64 consecutive rows)
  // (diag(Q_nn)-type row quantities without re-reading the n x m result: lib/fitc_gp.ml:222-223, :1048, :1164)
  // optional weighted column sums of the raw A operand, computed by the diagonal-tile blocks of an upper_only TN
  // launch while they stream A anyway:  cs_out[slice * N + c] = sum_{k in slice} A[k][c] * cs_w[k]
  // (c~ = V^T (y ./ s) without a separate pass over V; lib/fitc_gp.ml:285: gemv ~trans:`T q_mat y~)
  const T* cs_w = nullptr;
  double* cs_out = nullptr;
  double* rp_sumsq = nullptr;
  double* rp_dot = nullptr;
  const double* rp_vec = nullptr;
  // Round barrier (set by launch_gemm for the launches it pays on: fp32 SYRK-shaped split-K launches): the 64 workgroups
  // an XCD holds at a time -- one residency round of its item list: one 8 x 8 super tile of one k-slice -- wait for each
  // other before their first stage, so that they stream the 16 operand panels they share in step (round_ctr:
  // [8][round_stride] arrival counts, zeroed in front of the launch; a workgroup waits at most round_ticks x 10 ns).
  int* round_ctr = nullptr;
  int round_stride = 0;
  int round_ticks = 0;
  int lab_skip = 0;  // tools/gemm_check.hip only (timing ablation): 1 = no epilogue at all
  int lab_nostep = 0;  // engine lab (GPRHIP_LAB_NOSTEP=1, timing only, results wrong): the operand pointers do not advance
                       // from stage to stage, so every refill after the first hits the L2 -- how much of a launch is
                       // memory latency the two-buffer ring fails to hide
  unsigned long long* lab_ts = nullptr;  // tools/gemm_check.hip only: 8 words per workgroup (phase time stamps, HW_ID, XCC_ID)
};
using GemmArgs = GemmArgsT<double>;
using GemmArgsF = GemmArgsT<float>;

// Enqueue on `stream`.
void launch_gemm(GemmOp op, const GemmArgs& g, hipStream_t stream);
void launch_gemm(GemmOp op, const GemmArgsF& g, hipStream_t stream);

// A weighted upper_only TN launch with A == B (SYRK-shaped) split `kslices` ways (a multiple of 8) gives its diagonal
// tiles fewer, longer k-slices: their partial products (and the column sums) occupy slice buffers 0 .. this - 1 only.
int gemm_syrk_diag_slices(int kslices, bool f64, bool col_sums);

// Partial row sums the fused row reductions emit per 128-column tile (= wave columns of the engine geometry).
int gemm_row_parts_per_tile();

// One-time per-process setup (raises the dynamic-LDS limit of the kernels; fixes the engine geometry).
void gemm_init();

// Workgroups of the engine the device holds at once (two per CU; 512 on MI355X) -- read from the device by gemm_init().
// The split-K model of the mid-size SYRK launches (gprhip.hip, pick_kslices) takes its residency rounds from this.
int gemm_resident_slots();

}  // namespace gprhip
