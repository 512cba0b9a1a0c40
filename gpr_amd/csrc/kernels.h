// Device-kernel launchers of the FITC path other than the MFMA engine.
// Layout conventions (all fp64, device memory):
//   points   : "point-major" [n][dim]   (== the reference's Fortran dim x n Bigarray, lib/interfaces.ml:190-195)
//   inducing : [mp][d], rows >= m are padding
//   n x m matrices (K_nm and everything derived from it): row-major [rows_padded][mp],
//              one training point per row ("K_mn column-major" in BASELINE.json's words)
//   m x m matrices: row-major [mp][mp]; symmetric ones are valid in the upper triangle
//              (tile-granular: every 128x128 tile with row tile <= column tile is fully written)
#pragma once
#include "common.h"

namespace gprhip {

struct CovParams {
  int kind;             // 0 = Cov_se_iso, 1 = Cov_se_fat (projection-only sub-case)
  double log_sf2, sf2;  // lib/cov_se_iso.ml:41-44 / lib/cov_se_fat.ml:62-75
  double inv_ell2;      // iso: exp(-2 log_ell); fat: 1
  double inv_ell2_05;   // iso: -0.5*inv_ell2;   fat: -0.5
  const double* ms;     // Cov_se_fat multiscales exp(log_multiscales_m05)+0.5 as [mp][d] (padding 1), or null
};

// K_m (lib/cov_se_iso.ml:56-87, lib/cov_se_fat.ml:85-100): km = covariance (padding rows/cols 0),
// kj = km + (het + jitter) on the real diagonal (het = heteroskedastic noise, may be null,
// lib/cov_se_fat.ml:136-142) and exactly 1 on the padded diagonal.
void launch_cov_upper(const CovParams& cp, const double* Z, int m, int mp, int d, double jitter,
                      const double* het, double* km, double* kj, hipStream_t s);

// K_nm rows [0, rows) of a chunk (lib/cov_se_iso.ml:128-159, lib/cov_se_fat.ml:224-240);
// rows in [rows, rows_p) and columns in [m, mp) are written as 0.
// TS = storage type of the n x m matrices (double, or float for the fp32-bulk mode: distances and
// exp are still evaluated in fp64, only the stored value is rounded).
// shift (optional, [d]: centroid of the inducing points): lets the fp32-bulk mode with 16..64 point dimensions take
// the matrix-core distance kernel (cov_cross_mfma_kernel); fp64 storage always runs the direct-difference kernels.
template <typename TS>
void launch_cov_cross(const CovParams& cp, const double* pts, int rows, int rows_p, const double* Z,
                      int m, int mp, int d, TS* K, hipStream_t s, const double* shift = nullptr);

// Cov_se_fat.Eval.Inputs.project (lib/cov_se_fat.ml:215-218): P[n][d] = X[n][D] * tproj[D][d]
// (tproj given as the reference's Fortran D x d matrix, i.e. element (big,small) at tproj[small*D+big]).
void launch_project(const double* X, int64_t n, int D, int d, const double* tproj, double* P,
                    hipStream_t s);

// ---- blocked Cholesky / triangular inverse pieces (chol.hip)
// Factor the 128x128 diagonal block j of the row-major mp x mp matrix A in place (upper, A = U^T U),
// zero its strict lower part, and write inv(U_jj) (upper, lower zero) to dinv (128x128 row-major).
// On a non-positive pivot, *info (if still 0) is set to the 1-based global index.
void launch_potrf_diag(double* A, int mp, int j, double* dinv, int* info, hipStream_t s);
// flags bit0: the block already holds an upper factor -- skip the factorisation, only write inv(U_jj) to dinv
// m_real (0 = mp): rows and columns from m_real on are identity padding; micro-panels wholly inside it are skipped
void potrf_fetch_timestamps(unsigned long long* out64);  // tools/potrf_check: phase stamps of a flags-bit-8 launch
void launch_potrf_diag_flags(double* A, int mp, int j, double* dinv, int* info, int flags, hipStream_t s, int m_real = 0);
// Whole blocked factorisation A = U^T U (upper, in place; the strict lower parts of the diagonal blocks are zeroed, the
// tiles below the diagonal untouched) + inv(U_jj) of every diagonal block in dinv [mp/128][128][128], without the
// contraction engine: factor-only diagonal kernel, substitution panel, small-tile trailing update per 128-row step.
// With Xinv != null the inverse of the factor, inv(U) (upper, zeros below), is produced by the same steps (an identity
// right-hand side in the mp x mp scratch Yscratch rides along) and dinv only serves as scratch; with Xinv == null dinv
// receives the block inverses as before.
// aux (optional): a side stream and four events of the caller's -- the part of every step's trailing update that the next
// diagonal block does not need then runs on the side stream beside that block's factorisation (look-ahead).
struct PotrfAux {
  hipStream_t side = nullptr;
  int min_rest = 0;  // steps whose update has at least this many sub-tiles outside block row j + 1 use the side stream (0: none)
  hipEvent_t ev_panel[2] = {nullptr, nullptr}, ev_rest[2] = {nullptr, nullptr};
};
void potrf_upper_blocked(hipStream_t s, double* A, int mp, double* dinv, int* info, double* Yscratch = nullptr,
                         double* Xinv = nullptr, int m_real = 0, const PotrfAux* aux = nullptr);
// The same factorisation + inverse as ONE persistent launch with device-side dependencies (chol.hip, round 5): the
// workspace holds the task lists and flag words of an mp x mp problem (mp / 128 >= 2 blocks; create returns null
// otherwise) on the current device.  *info: first non-positive pivot (1-based), or POTRF_CHAIN_ABORT_CODE if a
// dependency wait ran into its bound (an internal error: the caller reports GPRHIP_EHIP).
constexpr int POTRF_CHAIN_ABORT_CODE = 0x7ffffff0;
#ifdef GPRHIP_LAB
struct PotrfChain;
PotrfChain* potrf_chain_create(int mp);
void potrf_chain_destroy(PotrfChain* ch);
// trace (device, [ntasks + mp/128][4] 64-bit words, or null): per task / diagonal block the 100 MHz wall-clock stamps
// {taken, dependencies seen, work done, flag set} -- tools/potrf_check TRACE=<m>
void potrf_upper_chain(hipStream_t s, PotrfChain* ch, double* A, int mp, double* dinv, int* info, double* Y, double* Xinv,
                       int m_real = 0, unsigned long long* trace = nullptr);
int potrf_chain_tasks(const PotrfChain* ch, int* kinds4);  // the task list ({kind, j, a, b} each), returns its length
#endif
// Single-block matrices (mp = 128): A = chol(I + src) (upper, in place of a load: src = the (0,0) tile of an exchange-1
// buffer), Xinv = A^-1, and the m-vectors that follow in pass 2 (chol.hip, potrf_diag_body<true>):
//   b = Xinv^T cvec, t~ = Xinv b, t = uinv t~, logdet = log|I + src|, bb = |b|^2
struct PotrfFuse {
  const double* src = nullptr;
  const double* cvec = nullptr;
  const double* uinv = nullptr;  // U^-1 [128][128]
  const double* tail_in = nullptr;  // four scalars copied to tail_out (the exchange-1 tail, into the result block)
  double* tail_out = nullptr;
  double *bvec = nullptr, *ttil = nullptr, *tvec = nullptr, *logdet = nullptr, *bb = nullptr;  // out
};
void launch_potrf_fused(const PotrfFuse& f, double* A, double* Xinv, int* info, int m_real, hipStream_t s);
// At most 64 inducing points of at most 16 dimensions, no multiscales: A = chol(K_m + het + jitter) with the matrix built
// in the kernel (launch_cov_upper's values: km = the covariance alone), Xinv = A^-1; both 128 x 128 with identity padding.
struct PotrfKm {
  CovParams cp{};
  const double* Z = nullptr;
  int m = 0, d = 0;
  double jitter = 0.0;
  const double* het = nullptr;
  double* km = nullptr;
};
void launch_potrf_km(const PotrfKm& g, double* A, double* Xinv, int* info, hipStream_t s);
void launch_zero_strict_lower(double* A, int mp, hipStream_t s);
void launch_copy_block(const double* src, int64_t lds, double* dst, int64_t ldd, int rows, int cols,
                       hipStream_t s);
// X (mp x mp) = block-diagonal matrix of the nb 128x128 blocks stored consecutively in dinv
void launch_scatter_diag_blocks(const double* dinv, int mp, double* X, hipStream_t s);

// ---- small dense vector ops on m-vectors (single block)
// y = op(A) x for an upper-triangular row-major mp x mp A; trans=0: y_i = sum_{k>=i} A[i][k] x_k,
// trans=1: y_i = sum_{k<=i} A[k][i] x_k.
void launch_triu_matvec(const double* A, int mp, const double* x, double* y, int trans, hipStream_t s);
// ... with one more workgroup that forms a scalar on the side: rider 1: rout = 2 sum_{i<rn} log rA[i (mp+1)]; 2: sum_{i<rn} rA[i]^2
void launch_triu_matvec_rider(const double* A, int mp, const double* x, double* y, int trans, int rider, const double* rA,
                              int rn, double* rout, hipStream_t s);

// ---- row kernels (rowops.hip)
struct Pass1RowArgs {
  const double* part;    // [npart][ld] partial row sums of V^2 from the epilogue of the V product (part-major);
                         //   null = keep r of the previous evaluation (update_sigma2)
  int npart;
  int64_t ld;
  const double* y;       // [rows] targets of this chunk (may be null: model-only)
  int rows;
  double sf2, sigma2;
  double* r;             // out [rows]
  double* is;            // out [rows_p] (padding rows get 0)
  double* yis;           // out [rows_p]  is*y (padding 0)
  double* partial;       // out [nblocks][4]: sum log s, sum is*y^2, sum is*r, unused
};
int pass1_row_blocks(int rows);
void launch_pass1_rows(const Pass1RowArgs& a, hipStream_t s);

struct Pass2RowArgs {
  const double* part_sq;  // [npart][ld] partial row sums of Q'^2 from the epilogue of the Q' product
  const double* part_dot; // [npart][ld] partial row sums of Q' .* b  (b = Rinv^T c = Q_n^T y~ of the reference)
  int npart;
  int64_t ld;
  const double* y;       // [rows] or null
  const double* is;      // [rows_p]
  const double* r;       // [rows]
  int rows, variational;
  double sf2;
  double* es;            // optional out [rows_p]: rowsum(X .* K) = q - v (sf2 - r) - w (K t)  (Proj gradient)
  double* w;             // out [rows_p]  (padding 0)
  double* v;             // out [rows_p]  (padding 0)
  double* partial;       // out [nblocks][4]: sum v, sum is, sum w*(y-Kt) (= sum is*res^2), sum v1
};
void launch_pass2_rows(const Pass2RowArgs& a, hipStream_t s);

// Per row of M [rows_p][mp]: sumsq[row] = sum_c M^2, dot[row] = sum_c M*b (either output may be null).
// Prediction path: Means.calc / Variances.calc (lib/fitc_gp.ml:418-425, :498-518).
template <typename TS>
void launch_row_sumsq_dot(const TS* M, const double* b, int rows, int mp, double* sumsq, double* dot,
                          hipStream_t s);
// var[r] = sf2 - k[r] + b[r] (+ sigma2 if predictive)   lib/fitc_gp.ml:509-517, :520-526
void launch_variance_combine(const double* k, const double* b, int rows, double sf2, double add,
                             double* var, hipStream_t s);

// dst = sum over split-K slices on a batched rows x cols rectangle (slices and dst share offsets)
void launch_sum_slices_rect(const double* slices, int nslices, int64_t stride, int rows, int cols, int64_t ld,
                            int nbatch, int64_t bs, double* dst, hipStream_t s);

// ---- posterior paths (posterior.hip)
// partial[block][4] = { sum (y-mean)^2, sum |y-mean|, max |y-mean|, sum y^2 } per 256 rows  (Stats, lib/fitc_gp.ml:304-374)
int residual_stat_blocks(int rows);
void launch_residual_stats(const double* y, const double* mean, int rows, double* partial, hipStream_t s);
// out (np x np row-major, symmetric, identity padding) from the upper triangle of a Fortran nt x nt matrix; diag += add
void launch_sym_from_upper(const double* in, int64_t ld, int nt, double* out, int np, double add, hipStream_t s);
// C[i][i] += (vec ? vec[i] : 0) + add
void launch_add_diag(double* C, int64_t ld, int n, const double* vec, double add, hipStream_t s);
// S[s][i] += v[i] for the ns rows of S
void launch_add_row_vector(double* S, int64_t ld, int ns, int n, const double* v, hipStream_t s);
// out[i] = a - x[i]
void launch_const_minus(const double* x, int n, double a, double* out, hipStream_t s);

// out[col] (+)= sum_slab partial[slab][col]
void launch_reduce_rows(const double* partial, int nslabs, int width, double* out, int accumulate,
                        hipStream_t s);

template <typename TS>
struct GradArgs {
  const TS* X;           // [rows_p][mp]  X of lib/fitc_gp.ml:1204-1206 for this chunk
  const double* pts;     // [rows][d]  inputs (iso) or projections (fat) of the chunk
  const double* Z;       // [mp][d]
  int rows, rows_p, m, mp, d;
  double log_sf2, inv_ell2_05;  // K_rc = exp(log_sf2 + inv_ell2_05*|x_r - z_c|^2) is recomputed on the fly
  const double* big;     // [rows][D] original inputs of the chunk (Cov_se_fat with tproj), else null
  int D;                 // big dimension (0 when big == null)
  double* colpart;       // out [nslabs][col_rows][mp]: row 0 = column sums of E, rows 1..d = sum_r p_kr E_rc,
                         //     rows d+1..d+D = sum_r x_big,r E_rc, then (multiscales) d rows sum_r p_kr^2 E_rc
  int col_rows;          // rows of one slab of colpart
  int slab;              // training points per slab (grad_slab_rows)
  double* scalpart;      // out [nslabs][nbx][2]: sum E, sum E*sqr_diff
  const double* shift;   // [d] common offset (centroid of the inducing points) the MFMA kernel subtracts from points
                         //     and inducing points before it expands |p - z|^2, or null
  const TS* K;           // [rows_p][mp] K_nm of the chunk kept from pass 1 (Cov_se_fat, matrix-core kernel), or null:
                         //     E = X .* K is then read instead of recomputed (no distance product, no exp)
  const double* ms;      // multiscales [mp][d] or null (Cov_se_fat)
  double* rowes;         // multiscales + tproj: out [rows][nslots][d] partial sum_c E_rc / ms_kc, nslots = 4*gridDim.x
};
// part[slab][big*d + small] = sum_{r in slab} x_big,r * p_small,r * es_r   (slab = 256 rows)
// es_ld == 1: one weight per row; es_ld == d: one per (row, small) (multiscales)
void launch_proj_term2(const double* X, const double* P, const double* es, int es_ld, int rows, int D, int d,
                       double* part, hipStream_t s);
// es[row][k] = sum_slot rowes[row][slot][k]
void launch_reduce_rowes(const double* rowes, int rows, int nslots, int d, double* es, hipStream_t s);
int grad_slab_rows(int col_rows);  // rows per slab of colpart / scalpart for a problem with that many accumulator rows
// d > 64 or D > 64: E = X .* K with K of the chunk given in memory (see rowops.hip)
template <typename TS>
void launch_grad_wide(const GradArgs<TS>& a, const TS* K, hipStream_t s);
// the same pass on the matrix cores (grad_mfma.hip): column blocks of 128, or 0 if the launch is not eligible
template <typename TS>
int grad_mfma_col_blocks(const GradArgs<TS>& a);
// with projection hypers the matrix-core kernel leaves rows 1..d of the column accumulators zero: after the slabs of
// all chunks are reduced, acc[1 + k] += sum_b tproj(b, k) acc[d + 1 + b]   (acc: [col_rows][mp])
void launch_proj_inducing_grad(double* acc, int mp, int d, int D, const double* tproj, hipStream_t s);
template <typename TS>
void launch_grad_mfma(const GradArgs<TS>& a, hipStream_t s);
template <typename TS>
void launch_grad_fused(const GradArgs<TS>& a, hipStream_t s);

// ---- m x m finalisation (finalize.hip)
// dst (upper tiles) = base + sum_z slices[z]; the diagonal tiles sum the first nslices_diag slices only (0 = nslices;
// gemm_syrk_diag_slices for the result of a SYRK-shaped engine launch)
template <typename TS>
void launch_sum_slices(const double* base, const TS* slices, int nslices, int64_t stride, int mp,
                       double* dst, hipStream_t s, int packed = 0, int nslices_diag = 0);
// dst (float) = src (double), n elements: fp32 copies of U^-1 / R~^-1 for the fp32 contractions
void launch_to_float(const double* src, float* dst, int64_t n, hipStream_t s);
// W~ = I - B~^-1 - t~ t~^T - G~ as a full symmetric matrix (inputs valid on upper tiles);
// the reference's W (lib/fitc_gp.ml:1196-1203) is U^-1 W~ U^-T.
// up to three device blocks into (device-visible) pinned host memory by a kernel (finalize.hip)
struct ShipArgs {
  const double* src[3] = {nullptr, nullptr, nullptr};
  double* dst[3] = {nullptr, nullptr, nullptr};
  int64_t n[3] = {0, 0, 0};
};
void launch_ship(const ShipArgs& a, hipStream_t s);
void launch_build_w(const double* binv, const double* t, const double* G, int mp, double* W,
                    hipStream_t s);
// Trace terms of W against K_m and its derivatives (lib/fitc_gp.ml:956-973, lib/utils.ml:196-220),
// as per-column partial sums over slabs of 256 rows: part[slab][q][c], q = 0: sum_r W_rc K_rc,
int km_slab_rows(int m);  // rows per slab of the km_traces partial buffers
// q = 1: sum_r W_rc K_rc |z_r - z_c|^2, q = 2+k: sum_r W_rc K_rc (z_kr - z_kc).  W, km full symmetric.
void launch_km_traces(const double* W, const double* km, const double* Z, int m, int mp, int d,
                      double* part, const CovParams& cp, hipStream_t s);
// Multiscale variant (lib/cov_se_fat.ml:441-516): part[slab][q][c] with q = 0: sum_r W_rc K_rc,
// q = 2+k: sum_{r!=c} W_rc K_rc (z_kr - z_kc)/(ms_kr + ms_kc - 1),
// q = 2+d+k: sum_{r!=c} W_rc K_rc (iscale - sdiff^2), iscale = 1/(ms_kr + ms_kc - 1), sdiff = (z_kr - z_kc) iscale
void launch_km_traces_ms(const double* W, const double* km, const double* Z, const double* ms, int m, int mp,
                         int d, double* part, hipStream_t s);

// ---- row passes of problems with few inducing points (small.hip): m <= 64, d <= 16 (8 with multiscales), D <= 64, fp64; the rows
// of all chunks are walked as one range (the resident stores are contiguous)
struct SmallPass1Args {
  CovParams cp;
  const double *pts, *Z, *uinv, *y;  // points [rows][d], inducing [mp][d], U^-1 [mp][mp], targets (or null)
  int rows, rows_p, m, mp, d;
  double sigma2;
  double *V, *r, *is, *yis;          // out: V [rows_p][mp] (padding zero), r / is / yis [rows_p]
  double* Kout;                      // out (or null): K_nm [rows_p][64] for pass 2
  double* part;                      // scratch, small_part_len doubles
};
struct SmallPass2Args {
  CovParams cp;
  const double *pts, *Z, *uinv, *rinv, *bvec, *ttil, *V, *y, *is, *r;
  const double* Kin;                 // K_nm [rows_p][64] as pass 1 left it, or null: recomputed (always for d > 8, multiscales)
  const double* big;                 // original inputs [rows][D] (Cov_se_fat with tproj) or null
  int D, rows, rows_p, m, mp, d, variational;
  double *w, *v, *es, *X;            // out: w, v [rows_p], es [rows_p] (or null), X [rows_p][mp] (columns < 64; or null)
  double* part;
};
bool small_path_fits(int m, int mp, int d, int D, int64_t rows, bool ms);
int64_t small_part_len(int d, int D);
// pass 1 + its reduction into the exchange-1 buffer: (0,0) tile, c~ [mp], scalar tail [4]
void launch_small_pass1(const SmallPass1Args& a, double* tile, double* cvec, double* tail, hipStream_t s);
// pass 2 + its reduction into the exchange-2 buffer, every entry of which is written: (0,0) tile, column block
// (col_rows x mp), `Proj term [D*d], tail [8]
void launch_small_pass2(const SmallPass2Args& a, int col_rows, double* tile, double* colblk, double* proj, double* tail,
                        hipStream_t s);
// the finish stage of a gradient evaluation (m x m work on the 64 x 64 corners) in one workgroup
struct SmallFinishArgs {
  const double *uinv, *rinv, *ttil, *km, *Z;
  const double* ms;         // multiscales [mp][d] or null
  const double* g;          // (0,0) tile of the reduced exchange-2 buffer: G~ = V^T diag(v) V
  int m, mp, d, km_rows;    // km_rows: rows of kmred to write (0: sum W.*K, 1: sum W.*K.*dist, 2+k: per dimension; with
                            //   multiscales 2+d+k as km_traces_ms_kernel)
  double *wmat, *kmred, *wdiag;  // out (wdiag may be null)
  const double* gather_from;     // n_gather doubles copied to ex (the exchange-2 tail behind the result block)
  int64_t n_gather;
  double* ex;
};
void launch_small_finish(const SmallFinishArgs& a, hipStream_t s);
// means and / or variances of a chunk of test points in one kernel (same limits, no multiscales)
struct SmallPredictArgs {
  CovParams cp;
  const double *pts, *Z, *uinv, *rinv, *tvec;  // test points [rows][d] (projected), inducing points, U^-1, R~^-1, t
  int rows, m, mp, d;
  double add;                                  // sigma2 for predictive variances, else 0
  double *means, *vars;                        // out [rows]; either may be null
};
void launch_small_predict(const SmallPredictArgs& a, hipStream_t s);

// ---- up to 256 inducing points that small.hip does not take (one or two 128-column tiles): one kernel per row pass + one
// reduction each, and the finish stage in two launches (mid.hip).  Same exchange-buffer layout as the engine path; the partial sums are per workgroup.
struct MidPass1Args {
  CovParams cp;
  const double *pts, *Z, *uinv, *y;  // points [rows][d], inducing [128][d], U^-1 [128][128], targets (or null)
  int rows, rows_p, m, mp, d;        // mp: 128 or 256
  double sigma2;
  double *V, *r, *is, *yis;          // out: V [rows_p][mp] (padding zero), r / is / yis [rows_p]
  double* part;                      // scratch, mid_part_len doubles
};
struct MidPass2Args {
  CovParams cp;
  const double *pts, *Z, *rinv, *bvec, *ttil, *V, *y, *is, *r;
  const double *uinvT, *rinvT;       // U^-T, R~^-T (launch_mid_transposes): the operands of the two "times B^T" products
  const double* big;                 // original inputs [rows][D] (Cov_se_fat with tproj) or null
  const double* shift;               // [>= d] centroid of the inducing points: expansion offset of the moments
  int D, rows, rows_p, m, mp, d, variational;
  double *w, *v, *es, *X;            // out: w, v [rows_p], es [rows_p] (or null), X [rows_p][mp] (or null)
  double* part;
};
struct MidFinishArgs {
  const double *uinv, *rinv, *ttil, *km, *Z;
  const double *uinvT, *rinvT;       // U^-T, R~^-T (launch_mid_transposes)
  const double* g;                   // packed upper tiles of the reduced exchange-2 buffer: G~ = V^T diag(v) V
  int m, mp, d, km_rows;                 // km_rows: rows of kmred to write (0: sum W.*K, 1: sum W.*K.*dist, 2+k: per dimension)
  double *wmat, *kmred, *wdiag;      // out (wdiag may be null)
  double* ybuf;                      // mp x mp scratch: Y = W~ U^-T between the two launches
  // result transfer from inside the second launch (null res_host: the caller copies): res_total doubles res_dev -> res_host
  // (pinned, device-addressable), four doubles a1_tail -> a1_host (or null), done_ctr a zeroed device int
  const double* res_dev = nullptr;
  double* res_host = nullptr;
  int64_t res_total = 0;
  const double* a1_tail = nullptr;
  double* a1_host = nullptr;
  int* done_ctr = nullptr;
  const double* gather_from;         // n_gather doubles copied to ex (the exchange-2 tail behind the result block)
  int64_t n_gather;
  double* ex;
};
constexpr int64_t MID_GRAM_ROWS = 4096;  // most rows of a shard whose two-tile Gram accumulations mid.hip's own launch pair takes (mid.hip: mid_gram_kernel)
constexpr int64_t MID_ROWS_TWO_TILES = 32768;  // most rows of a shard the two-tile kernels take (mid.hip: mid_path_fits)
bool mid_path_fits(int m, int mp, int d, int D, int64_t rows, bool ms);
int64_t mid_part_len(int mp, int d, int D);
void launch_mid_pass1(const MidPass1Args& a, double* tile, double* cvec, double* tail, hipStream_t s);
void launch_mid_pass2(const MidPass2Args& a, int col_rows, double* tile, double* colblk, double* proj, double* tail,
                      hipStream_t s);
void launch_mid_finish(const MidFinishArgs& a, hipStream_t s);
// Two tiles: the Gram accumulation V^T diag(w) V (upper 128-tiles, packed as the exchange buffers hold them) and, with y,
// the weighted column sums V^T y, over a shard's resident V (rows x 256) -- mid.hip's own launch pair for shards the
// engine's SYRK-shaped launch is too heavy for (its fixed cost is 28 us whatever the row count)
struct MidGramArgs {
  const double* V;                   // [rows_alloc][256]
  const double* w;                   // row weights (1/s in pass 1, v in pass 2)
  const double* y;                   // column-sum weights (y/s) or null
  int rows;                          // real rows of the shard
  double* part;                      // scratch, mid_gram_part_len() doubles
};
int64_t mid_gram_part_len();
void launch_mid_gram(const MidGramArgs& a, double* tiles, double* cvec, hipStream_t s);
void launch_mid_transposes(const double* uinv, const double* rinv, int mp, double* uinvT, double* rinvT, hipStream_t s);

}  // namespace gprhip
