// C ABI + orchestration of the two-pass streaming FITC evaluation (see include/gprhip.h).
//
// Whitened formulation (V = K_nm U^-1 with U = chol(K_m + jitter)): every quantity of the reference's
// stacked-QR path is obtained from  B~ = I + V^T diag(1/s) V = R~^T R~  -- no K_m^-1 / B^-1 cancellation,
// QR-grade accuracy from SYRK + potrf.
// Pass 1 (per row chunk):  K = cov(X, Z);  V = K U^-1 (kept in HBM for pass 2);  r, s, 1/s;
//                          B~_part += V^T diag(1/s) V;  c~ += V^T (y/s).
// Exchange 1            :  sum over shards of (B~_part, c~, sum log s, sum y^2/s, sum r/s).
// Middle (m x m)        :  R~ = chol(I + B~_part);  b = R~^-T c~;  t~ = R~^-1 b;  t = U^-1 t~;  R~^-1.
//                          l1, l2 are complete here: evidence-only evaluations stop after this.
// Pass 2 (per row chunk):  Q' = V R~^-1;  q_diag, w, v;  X~ = diag(1/s) Q' R~^-T - diag(v) V - w t~^T
//                          (fused GEMM epilogue);  X = X~ U^-T;  G~_part += V^T diag(v) V;
//                          fused gradient accumulators over E = X .* K_nm.
// Exchange 2            :  sum over shards of (G~_part, gradient column accumulators, scalars).
// Finish                :  W = U^-1 (I - B~^-1 - t~ t~^T - G~) U^-T;  traces;  dl/dsigma2, dl/dtheta.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>
#include <string>
#include "../../include/gprhip.h"
#include "common.h"
#include "kernels.h"
#include "mfma_gemm.h"
#include "problem_internal.h"

namespace gprhip {
const std::string& last_error();

// x <- x / |x| over the first n entries (single block); nrm[0] = |x| before the scaling
__global__ __launch_bounds__(256) void normalize_kernel(double* __restrict__ x, int n, double* __restrict__ nrm) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += x[i] * x[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  const double r = sqrt(red[0]);
  const double inv = r > 0.0 ? 1.0 / r : 0.0;
  for (int i = threadIdx.x; i < n; i += 256) x[i] *= inv;
  if (threadIdx.x == 0) nrm[0] = r;
}

__global__ void copy_diag_kernel(const double* __restrict__ A, int mp, double* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < mp) out[i] = A[(int64_t)i * (mp + 1)];
}

// dst = I + a on upper tiles, 0 elsewhere; a: packed upper tiles (the exchange-1 buffer)
__global__ void add_identity_upper_kernel(const double* __restrict__ a, int mp, double* __restrict__ dst) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int r = blockIdx.y;
  if (c >= mp) return;
  const int64_t off = (int64_t)r * mp + c;
  dst[off] = (r / TILE <= c / TILE) ? a[packed_upper_off(r, c)] + (r == c ? 1.0 : 0.0) : 0.0;
}

}  // namespace gprhip

using namespace gprhip;

extern "C" int64_t gprhip_ar1_len(const gprhip_problem* p);
extern "C" int64_t gprhip_ar2_len(const gprhip_problem* p);

namespace {

constexpr double LOG_2PI = 1.8378770664093454835606594728112;  // lib/utils.ml:39-40
constexpr int NSCAL = 16;
enum {  // device scalar slots
  SC_LOGDET_B = 0,  // log |B~| = log |B| - log |K_m + jitter|
  SC_BB = 1,        // |b|^2 = |Q_n^T y~|^2
  SC_LMAX = 2,      // power-iteration estimates of lambda_max(K_m + jitter) and lambda_max((K_m + jitter)^-1)
  SC_LMAX_INV = 3,
  SC_A1TAIL = 8,    // 8..11: copy of the reduced exchange-1 tail (single-block problems: written by the fused B~ phase)
};
// tail of the exchange-1 buffer
enum { A1_SUMLOGS = 0, A1_ISY2 = 1, A1_ISR = 2, A1_TAIL = 4 };
// tail of the exchange-2 buffer
enum { A2_SUMV = 0, A2_SUMIS = 1, A2_WRES = 2, A2_SUMV1 = 3, A2_SUME = 4, A2_SUMED = 5, A2_TAIL = 8 };

struct Timer {
  std::vector<std::pair<std::string, std::pair<hipEvent_t, hipEvent_t>>> ev;
  bool on = false;       // level 2: an event pair around every stage of the evaluation
  bool kernel = false;   // level 1: one event pair around the dominant kernel alone (the pass-1 SYRK launch)
  hipEvent_t k0 = nullptr, k1 = nullptr;
  bool k_recorded = false;
};

}  // namespace

struct gprhip_problem {
  int device = 0, kind = 0;
  int64_t n = 0;
  int D = 0, d = 0, m = 0, mp = 0;
  int64_t chunk = 0;
  int nchunks = 0;
  int kslices = 0;   // upper bound on the split-K factor (set at creation from the memory budget)
  hipStream_t stream = nullptr;
  // second stream: the covariance of the first row chunk runs beside the K_m factorisation (do_pass1)
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipEvent_t ev_rf = nullptr, ev_binv = nullptr;  // pass 2: R^-1 and B~^-1 come from the second stream (do_pass2)
  // GPRHIP_COV_OVERLAP (read at creation): the covariance of row chunk c + 1 is built on the second stream beside the
  // V = K U^-1 product of chunk c (the two chunk buffers alternate in pass 1) instead of in front of its own product
  int cov_overlap = 0;
  hipEvent_t ev_cov[2] = {nullptr, nullptr}, ev_vdone[2] = {nullptr, nullptr};
  // third stream + events: look-ahead of the blocked factorisation (chol.hip, potrf_upper_blocked) -- a stream of its own,
  // because stream2 carries the first row chunk's covariance beside the K_m factorisation
  PotrfAux potrf_aux{};
  std::vector<void*> allocs;
  int64_t alloc_bytes = 0, planned_bytes = 0;  // hipMalloc'ed so far / gprhip_memory_plan's total for this problem

  double *X = nullptr, *y = nullptr, *P = nullptr;
  double *Z = nullptr, *tproj = nullptr;
  double *km = nullptr, *kj = nullptr, *umat = nullptr, *uinv = nullptr;
  double *bmat = nullptr, *rinv = nullptr, *binv = nullptr, *wtil = nullptr, *wmat = nullptr,
         *tmp = nullptr, *dinv = nullptr;
  double *bvec = nullptr, *ttil = nullptr, *tvec = nullptr, *scal = nullptr;
  int* info = nullptr;
  double *r = nullptr, *is = nullptr, *yis = nullptr, *w = nullptr, *v = nullptr, *es = nullptr;
  double* projpart = nullptr;
  double* zshift = nullptr;  // [64] centroid of the inducing points (gradient kernel's expansion offset)
  double *rp1 = nullptr, *rp2 = nullptr;  // per-row partial sums from the GEMM epilogues [parts*mp/128][chunk]
  double *xt = nullptr, *pt = nullptr, *prow = nullptr;  // prediction: test-point chunk, its projection, 3 row vectors
  int64_t xt_rows = 0;                                   // rows they hold
  void *predA = nullptr, *predB = nullptr;               // prediction chunk buffers beyond the training chunk (do_predict)
  int64_t pred_rows = 0;
  bool have_model = false;
  bool have_factors = false;  // U^-1 / R~^-1 valid (false after a means-only gprhip_load_predictor)
  bool have_v = false;        // Vstore / r hold V = K_nm U^-1 of the current kernel and inducing points (reuse_v)
  bool merged_x = false;      // this evaluation forms X by the two-phase product (set in pass 2)
  double cond_km = -1.0;       // 2-norm condition estimate of K_m + jitter of the last evaluation (< 0: not computed yet)
  double f32_coeff_tol = 0.25; // GPRHIP_F32_COEFF_TOL (read at creation): fp32-bulk problems refuse to use their mean
                               // coefficients when cond_km * 2^-24 exceeds it (0 = never refuse).  Measured over d = 1..16,
                               // m = 50..512 (profiles/r04_f32_guard.txt): the coefficients' error is 1/17 .. 1/3600 of that
                               // worst-case bound, i.e. <= ~1.5e-2 at the default threshold (cond ~ 4e6), typically 4e-3
  const void* x_last = nullptr;  // single-chunk gradient evaluations: the chunk buffer that holds X (debug fetch "x_rows")
  bool have_k = false;        // Kstore holds K_nm of the current kernel and inducing points for every chunk
  int k_resident = 1;         // GPRHIP_K_RESIDENT=0 (read at creation): never keep K_nm (ablation)
  // GPRHIP_SMALL_PATH=0 (read at creation): never take the one-kernel passes for at most 64 inducing points (small.hip)
  int small_path = 1;
  // GPRHIP_MID_PATH=0 (read at creation): never take the one-kernel passes for 65 .. 128 inducing points (mid.hip)
  int mid_path = 1;
  double* mid_part = nullptr;    // their per-workgroup partial sums (allocated at first use)
  int mid_gram = 1;              // GPRHIP_MID_GRAM: two tiles -- the Gram accumulations by mid.hip's own launch pair (0: the engine's)
  double* mid_gram_part = nullptr;  // its per-slice partial tiles (allocated at first use)
  int* mid_done = nullptr;       // arrival counter of their last finish launch (its last workgroup ships the results)
  double* small_part = nullptr;  // their per-workgroup partial sums (allocated at first use)
  double* small_k = nullptr;     // K_nm [rows_p][64] of the small pass 1, read back by the small pass 2
  bool small_k_valid = false;    // ... of the current kernel parameters and inducing points (as have_v)
  int w_as_ws = 1;            // GPRHIP_W_AS_WS=0 (read at creation): pass-2 SYRK through the plain weighted kernel (do_pass2)
  // GPRHIP_MERGED_X (read at creation): 0 = X~ and X = X~ U^-T as two launches, as in rounds 1-2 (A/B runs); 1 (default) =
  // the two-phase product for shards large enough to pay for R^-1; 2 = always (parity tests at small sizes)
  int merged_x_mode = 1;
  // GPRHIP_POTRF_ENGINE=1 (read at creation): the round-2 factorisation (engine launches per step) + recursive-doubling inverse
  bool engine_steps = false;
  // GPRHIP_POTRF_CHAIN=1 (read at creation; default 0): factorisation + inverse of matrices of two or more 128-blocks as
  // ONE persistent launch with device-side dependencies (chol.hip, potrf_upper_chain) instead of three launches per step.
  // Built in round 5, bit-identical, and measured 1.4 - 2.8 x SLOWER than the launches: kept for A/B
  int potrf_chain_mode = 0;
#ifdef GPRHIP_LAB
  PotrfChain* chain = nullptr;  // its task list and flag words (created at the first factorisation)
#endif
  // n x m storage: double, or float in the fp32-bulk mode (element size `esz`)
  void *bufA = nullptr, *bufB = nullptr, *Vstore = nullptr;
  // Cov_se_fat with projection hypers: K_nm of all rows, kept from pass 1 for the gradient kernel of pass 2 (which then
  // reads E = X .* K instead of recomputing distances and exp) when the device has room for a second n x m matrix
  void* Kstore = nullptr;
  bool kstore_tried = false;
  float *uinv_f = nullptr, *rinv_f = nullptr;  // fp32 copies of U^-1 / R~^-1 (fp32-bulk mode)
  double* rfinv = nullptr;   // R^-1 = U^-1 R~^-1 (R = R~ U is the reference's r_mat): B operand of the two-phase X product
  float* rfinv_f = nullptr;
  float *is_f = nullptr, *yis_f = nullptr, *v_f = nullptr;  // fp32 copies of the SYRK row weights (fp32-bulk mode)
  int f32 = 0;
  size_t esz = 8;
  void* slices = nullptr;
  int64_t slices_bytes = 0;
  double *rowpart = nullptr, *gemvpart = nullptr, *colpart = nullptr, *scalpart = nullptr,
         *kmpart = nullptr, *kmred = nullptr;
  double *ar1 = nullptr, *ar2 = nullptr;  // internal exchange buffers for single-device eval

  // state of the evaluation in flight
  gprhip_hypers h{};
  CovParams cp{};
  int want_grad = 0;
  int64_t n_total = 0;
  int stage = 0;  // 0 idle, 1 pass1 done, 2 pass2 done
  bool have_inputs = false, have_targets = false;
  // Per-evaluation parameters cross the PCIe link as ONE block: [Z (mp x d) | centroid (64) | tproj (D x d) | het (mp) |
  // multiscales (mp x d)] (the last three for Cov_se_fat), assembled in pinned host memory (hy_host, whose parts the
  // gradient assembly reads again) and copied by one asynchronous transfer of the prefix in use.
  double *hy_dev = nullptr, *hy_host = nullptr;
  int64_t hy_len = 0;
  hipEvent_t ev_hy = nullptr;  // the last upload of hy_host has been consumed
  double *hShift = nullptr, *hZ = nullptr, *hTproj = nullptr, *hHet = nullptr, *hMs = nullptr;  // parts of hy_host
  double* het = nullptr;                  // device copy of hHet
  // Results of an evaluation come back as one block as well: [scalars (NSCAL) | potrf flags (2 ints in one double) |
  // t (mp) | K_m traces (km_rows x mp) | diag W (mp)] = res_dev -> res_host (pinned), plus the tails of the two exchange
  // buffers (ex_host: [A1_TAIL | column block .. end of the exchange-2 buffer]).
  // ex_host / ex_dev lie directly behind the result block (one transfer can bring both).  Since round 6 the transfer is a
  // kernel's stores into res_host (mapped pinned memory) wherever it would exceed 32 KB or take more than one copy:
  // ship_kernel (finalize.hip) on the engine path, the last workgroup of mid_finish2_kernel (mid.hip)
  double *res_dev = nullptr, *res_host = nullptr, *ex_host = nullptr, *ex_dev = nullptr;
  bool a1_in_scal = false;  // this evaluation's exchange-1 tail is in the result block's scalars (SC_A1TAIL)
  int64_t res_len = 0, ex_len = 0;
  double *ms = nullptr, *rowes = nullptr, *es2 = nullptr;  // multiscales [mp][d]; per-row E/ms partials
  double* wdiag = nullptr;  // diag W inside the result block
  Timer timer;
  std::vector<std::string> tnames;
  std::vector<float> tms;

  template <typename T>
  T* alloc(int64_t count) {
    void* ptr = nullptr;
    GPR_HIP(hipMalloc(&ptr, (size_t)std::max<int64_t>(count, 1) * sizeof(T)));
    allocs.push_back(ptr);
    alloc_bytes += std::max<int64_t>(count, 1) * (int64_t)sizeof(T);
    return static_cast<T*>(ptr);
  }
  const double* pts() const { return kind == GPRHIP_COV_SE_FAT && h.tproj ? P : X; }
  int64_t rows_of(int c) const { return std::min<int64_t>(chunk, n - (int64_t)c * chunk); }
  // rows of the resident n x m store: full chunks are contiguous, the last one is padded to the tile
  int64_t rows_total_padded() const {
    return (int64_t)(nchunks - 1) * chunk + round_up(rows_of(nchunks - 1), TILE);
  }
  int ks_used = 8;
  int64_t slice_rows = 8192;  // training points per split-K slice of the SYRK launches (both precisions)
  // training points per workgroup of the gradient kernels (grad_slab_rows: 256 or 1024), halved down to 64 while the launch
  // would leave most of the chip idle (n = 2000, m = 128: 8 workgroups of 256 rows took 30 us; set at creation)
  int grad_slab = 256;
  int tile_order = 3;  // block -> tile order of the chunk GEMMs (mfma_gemm.hip tile_of_block): paired column tiles, XCD-local groups
  int grad_scalar = 0;  // GPRHIP_GRAD_SCALAR: use the scalar gradient kernel even where the MFMA one applies
  // Cov_se_fat `Proj hypers: rows of the exchange-2 column block beyond d+1, and the D x d second term
  int dbig() const { return kind == GPRHIP_COV_SE_FAT ? D : 0; }
  bool has_proj() const { return kind == GPRHIP_COV_SE_FAT && h.tproj != nullptr; }
  bool has_het() const { return kind == GPRHIP_COV_SE_FAT && h.log_hetero_skedasticity != nullptr; }
  bool has_ms() const { return kind == GPRHIP_COV_SE_FAT && h.log_multiscales_m05 != nullptr; }
  int64_t km_rows() const { return d + 2 + (kind == GPRHIP_COV_SE_FAT ? d : 0); }
  // exchange-2 column block: sum E, sum p_k E (d), sum x_big E (D), and for Cov_se_fat sum p_k^2 E (d)
  int64_t col_rows() const { return d + 1 + dbig() + (kind == GPRHIP_COV_SE_FAT ? d : 0); }
  bool use_small() const {
    return small_path && !f32 && !engine_steps && small_path_fits(m, mp, d, has_proj() ? D : 0, n, has_ms());
  }
  // one or two 128-column tiles of inducing points that the small path does not take (65 .. 256 of them, or fewer with more
  // input dimensions than small.hip stages): the row passes and the finish stage of mid.hip
  bool use_mid_gram() const { return mid_gram && n <= MID_GRAM_ROWS; }
  bool use_mid() const {
    return mid_path && !f32 && !engine_steps && !use_small() && mid_path_fits(m, mp, d, has_proj() ? D : 0, n, has_ms());
  }
};

namespace {

template <typename TS> const TS* inv_u(const gprhip_problem* p);
template <> const double* inv_u<double>(const gprhip_problem* p) { return p->uinv; }
template <> const float* inv_u<float>(const gprhip_problem* p) { return p->uinv_f; }
// per-row weights of the SYRK-shaped launches in the engine's element type
template <typename TS> const TS* row_weights(gprhip_problem* p, const double* w, float* wf);
template <> const double* row_weights<double>(gprhip_problem*, const double* w, float*) { return w; }
template <> const float* row_weights<float>(gprhip_problem* p, const double* w, float* wf) {
  launch_to_float(w, wf, (int64_t)p->nchunks * p->chunk, p->stream);
  return wf;
}
template <typename TS> const TS* inv_r(const gprhip_problem* p);
template <> const double* inv_r<double>(const gprhip_problem* p) { return p->rinv; }
template <> const float* inv_r<float>(const gprhip_problem* p) { return p->rinv_f; }
template <typename TS> const TS* inv_rfull(const gprhip_problem* p);
template <> const double* inv_rfull<double>(const gprhip_problem* p) { return p->rfinv; }
template <> const float* inv_rfull<float>(const gprhip_problem* p) { return p->rfinv_f; }

void need_trustworthy_coeffs(gprhip_problem* p, const char* who);

void tstart(gprhip_problem* p, const char* name) {
  if (!p->timer.on) return;
  hipEvent_t a, b;
  GPR_HIP(hipEventCreate(&a));
  GPR_HIP(hipEventCreate(&b));
  GPR_HIP(hipEventRecord(a, p->stream));
  p->timer.ev.push_back({name, {a, b}});
}
void tstop(gprhip_problem* p) {
  if (!p->timer.on) return;
  GPR_HIP(hipEventRecord(p->timer.ev.back().second.second, p->stream));
}
void tcollect(gprhip_problem* p) {
  p->tnames.clear();
  p->tms.clear();
  if (p->timer.k_recorded) {
    float ms = 0;
    hipEventElapsedTime(&ms, p->timer.k0, p->timer.k1);
    p->tnames.push_back("kernel_p1_syrk_B");
    p->tms.push_back(ms);
    p->timer.k_recorded = false;
  }
  for (auto& e : p->timer.ev) {
    float ms = 0;
    hipEventElapsedTime(&ms, e.second.first, e.second.second);
    bool found = false;
    for (size_t i = 0; i < p->tnames.size(); ++i)
      if (p->tnames[i] == e.first) {
        p->tms[i] += ms;
        found = true;
      }
    if (!found) {
      p->tnames.push_back(e.first);
      p->tms.push_back(ms);
    }
    hipEventDestroy(e.second.first);
    hipEventDestroy(e.second.second);
  }
  p->timer.ev.clear();
}

// Blocked upper Cholesky A = U^T U in place (dpotrf `U; lib/fitc_gp.ml:56) -- diagonal blocks in
// LDS, panel solve and trailing update on the MFMA engine.  dinv receives inv(U_jj) per block.
void potrf_upper_n(hipStream_t s, double* A, int mp, double* dinv, int* info, bool engine_steps, int m_real = 0) {
  // default: the engine-free step kernels of chol.hip (potrf_upper_blocked); engine_steps (GPRHIP_POTRF_ENGINE=1 at
  // problem creation) keeps the round-2 sequence below for A/B timing -- diagonal block with its full inverse, panel
  // and trailing update as engine launches
  if (!engine_steps) {
    potrf_upper_blocked(s, A, mp, dinv, info, nullptr, nullptr, m_real);
    return;
  }
  const int nb = mp / TILE;
  for (int j = 0; j < nb; ++j) {
    double* dj = dinv + (int64_t)j * TILE * TILE;
    launch_potrf_diag(A, mp, j, dj, info, s);
    if (j + 1 < nb) {
      const int rest = (nb - 1 - j) * TILE;
      double* panel = A + (int64_t)j * TILE * mp + (int64_t)(j + 1) * TILE;
      GemmArgs g;  // panel <- inv(U_jj)^T * panel   (in place: a block reads its whole column tile first)
      g.A = dj; g.lda = TILE; g.B = panel; g.ldb = mp; g.C = panel; g.ldc = mp;
      g.M = TILE; g.N = rest; g.K = TILE;
      launch_gemm(OP_TN, g, s);
      GemmArgs u;  // trailing <- trailing - panel^T panel  (upper tiles)
      u.A = panel; u.lda = mp; u.B = panel; u.ldb = mp;
      u.C = A + (int64_t)(j + 1) * TILE * mp + (int64_t)(j + 1) * TILE; u.ldc = mp;
      u.M = rest; u.N = rest; u.K = TILE; u.alpha = -1.0; u.beta = 1.0; u.upper_only = 1;
      launch_gemm(OP_TN, u, s);
    }
  }
}
void potrf_upper(gprhip_problem* p, double* A, int* info) {
  potrf_upper_n(p->stream, A, p->mp, p->dinv, info, p->engine_steps, p->m);
}

// k-slices of an m x m product of triangular factors: few output tiles, each up to m / 128 blocks deep.  From three
// 128-blocks on the k-range is split (slices end on block boundaries) -- at m = 512 the two products of the finish stage
// took 64 and 69 us as 16 workgroups of up to four blocks each (profiles/r05_timeline_n50000_m512.txt).
static int mxm_slices(int mp) { return mp >= 3 * TILE ? std::min(4, mp / TILE) : 1; }

// A non-batched m x m product with few output tiles and a long k-range, split over `ks` k-slices so that the
// launch fills the chip; partial products go to the split-K scratch and are summed in a fixed order.
void gemm_splitk(gprhip_problem* p, GemmOp op, GemmArgs g, int ks) {
  const int64_t mm = (int64_t)p->mp * p->mp;
  if (ks < 2 || (int64_t)ks * mm * 8 > p->slices_bytes || g.ldc != p->mp) {
    launch_gemm(op, g, p->stream);
    return;
  }
  double* const dst = g.C;
  g.C = static_cast<double*>(p->slices);
  g.kslices = ks;
  g.slice_stride = mm;
  launch_gemm(op, g, p->stream);
  launch_sum_slices_rect(static_cast<double*>(p->slices), ks, mm, g.M, g.N, p->mp, 1, 0, dst, p->stream);
}

// inv(U) for the upper-triangular factor by recursive doubling over the 128-blocks:
//   inv([U11 U12; 0 U22]) = [X11, -X11 U12 X22; 0, X22]
// Level s joins neighbouring inverted diagonal blocks of s rows; all joins of a level are independent
// and run as one batched launch (two per level: T = U12 X22, X12 = -X11 T), so the whole inverse is
// 2 log2(mp/128) launches instead of one short GEMM chain per block column.
void trtri_upper(gprhip_problem* p, const double* U, double* X, double* tmp) {
  const int mp = p->mp;
  hipStream_t s = p->stream;
  launch_scatter_diag_blocks(p->dinv, mp, X, s);
  for (int64_t sz = TILE; sz < mp; sz *= 2) {
    const int64_t pair = 2 * sz;
    const int nfull = (int)(mp / pair);                 // pairs with two full halves
    const int64_t rem = mp - (int64_t)nfull * pair;     // a trailing pair with a short second half?
    auto join = [&](int64_t off, int64_t s2, int nb) {
      // The late levels have few output tiles and long k-ranges: split k over up to 8 slices so the launch fills
      // the chip (partial products go to the split-K scratch at the destination's offsets, then a fixed-order sum)
      const int64_t blocks = (int64_t)nb * (sz / TILE) * (s2 / TILE);
      auto slices_for = [&](int64_t kdim) {
        int ks = 1;
        while (ks < 8 && blocks * ks * 2 <= 256 && kdim / (ks * 2) >= 256) ks *= 2;
        return (double*)p->slices && (int64_t)ks * mp * mp * 8 <= p->slices_bytes ? ks : 1;
      };
      double* const sl = static_cast<double*>(p->slices);
      // T = U12 * X22   (X22 upper triangular)
      GemmArgs a;
      a.A = U + off * mp + off + sz; a.lda = mp; a.B = X + (off + sz) * (mp + 1); a.ldb = mp;
      a.C = tmp + off * mp + off + sz; a.ldc = mp;
      a.M = (int)sz; a.N = (int)s2; a.K = (int)s2; a.tri = TRI_KHI_BN;
      a.nbatch = nb; a.batch_a = a.batch_b = a.batch_c = pair * (mp + 1);
      const int ksa = slices_for(s2);
      if (ksa > 1) {
        a.C = sl + off * mp + off + sz; a.kslices = ksa; a.slice_stride = (int64_t)mp * mp;
        launch_gemm(OP_NN, a, s);
        launch_sum_slices_rect(sl + off * mp + off + sz, ksa, (int64_t)mp * mp, (int)sz, (int)s2, mp, nb,
                               pair * (mp + 1), tmp + off * mp + off + sz, s);
      } else {
        launch_gemm(OP_NN, a, s);
      }
      // X12 = -X11 * T  (X11 upper triangular)
      GemmArgs b;
      b.A = X + off * (mp + 1); b.lda = mp; b.B = tmp + off * mp + off + sz; b.ldb = mp;
      b.C = X + off * mp + off + sz; b.ldc = mp;
      b.M = (int)sz; b.N = (int)s2; b.K = (int)sz; b.tri = TRI_KLO_BM; b.alpha = -1.0;
      b.nbatch = nb; b.batch_a = b.batch_b = b.batch_c = pair * (mp + 1);
      const int ksb = slices_for(sz);
      if (ksb > 1) {
        b.C = sl + off * mp + off + sz; b.kslices = ksb; b.slice_stride = (int64_t)mp * mp;
        launch_gemm(OP_NN, b, s);
        launch_sum_slices_rect(sl + off * mp + off + sz, ksb, (int64_t)mp * mp, (int)sz, (int)s2, mp, nb,
                               pair * (mp + 1), X + off * mp + off + sz, s);
      } else {
        launch_gemm(OP_NN, b, s);
      }
    };
    if (nfull > 0) join(0, sz, nfull);
    if (rem > sz) join((int64_t)nfull * pair, rem - sz, 1);
  }
}

// U = chol(A) in place and X = inv(U): one pass of the step kernels with the identity riding along as right-hand side
// (chol.hip); `tmp` is an mp x mp scratch
void potrf_trtri(gprhip_problem* p, double* A, double* X, double* tmp, int* info) {
  if (p->engine_steps) {
    potrf_upper(p, A, info);
    trtri_upper(p, A, X, tmp);
    return;
  }
#ifdef GPRHIP_LAB
  if (p->potrf_chain_mode && p->mp >= 2 * TILE) {
    if (!p->chain) p->chain = potrf_chain_create(p->mp);
    if (p->chain) {
      potrf_upper_chain(p->stream, p->chain, A, p->mp, p->dinv, info, tmp, X, p->m);
      return;
    }
  }
#endif
  potrf_upper_blocked(p->stream, A, p->mp, p->dinv, info, tmp, X, p->m, p->potrf_aux.side ? &p->potrf_aux : nullptr);
}

// C (upper tiles) = X X^T for upper-triangular X: (U^T U)^-1 = U^-1 U^-T   (Utils.ichol, lib/utils.ml:110-113)
void triu_xxt(gprhip_problem* p, const double* X, double* C, hipStream_t st) {
  GemmArgs g;
  g.A = X; g.lda = p->mp; g.B = X; g.ldb = p->mp; g.C = C; g.ldc = p->mp;
  g.M = p->mp; g.N = p->mp; g.K = p->mp; g.tri = TRI_KLO_MAX; g.upper_only = 1;
  // few tiles, long triangular k-ranges: four k-slices (ending on block boundaries) fill the chip
  const int64_t mm = (int64_t)p->mp * p->mp;
  const int ks = (mxm_slices(p->mp) > 1 && 4 * mm * 8 <= p->slices_bytes) ? mxm_slices(p->mp) : 1;
  if (ks > 1) {
    g.C = static_cast<double*>(p->slices);
    g.kslices = ks;
    g.slice_stride = mm;
    launch_gemm(OP_NT, g, st);
    launch_sum_slices<double>(nullptr, static_cast<double*>(p->slices), ks, mm, p->mp, C, st);
  } else {
    launch_gemm(OP_NT, g, st);
  }
}

void upload_hypers(gprhip_problem* p, const gprhip_hypers* h) {
  if (!h || !h->inducing) {
    set_error("gprhip: hypers/inducing pointer is NULL");
    throw HipFail{ST_BAD_ARG};
  }
  if (h->sigma2 < 0.0) {
    set_error("Model.check_sigma2: sigma2 < 0");  // lib/fitc_gp.ml:148-149
    throw HipFail{ST_BAD_ARG};
  }
  if (p->kind == GPRHIP_COV_SE_ISO && h->tproj) {
    set_error("gprhip: tproj given for Cov_se_iso");
    throw HipFail{ST_BAD_ARG};
  }
  if (p->kind == GPRHIP_COV_SE_FAT && !h->tproj && p->D != p->d) {
    set_error("gprhip: Cov_se_fat without tproj needs D == d");
    throw HipFail{ST_BAD_ARG};
  }
  if (p->kind == GPRHIP_COV_SE_ISO && h->log_hetero_skedasticity) {
    set_error("gprhip: log_hetero_skedasticity given for Cov_se_iso");
    throw HipFail{ST_BAD_ARG};
  }
  if (h->log_multiscales_m05 && p->kind == GPRHIP_COV_SE_ISO) {
    set_error("gprhip: log_multiscales_m05 given for Cov_se_iso");
    throw HipFail{ST_BAD_ARG};
  }
  p->h = *h;                // (after every argument check: a refused call leaves no borrowed pointer behind)
  p->h.inducing = nullptr;  // borrowed; the padded copy lives in hZ
  // the optional arrays are presence flags from here on, pointing at the library's own pinned copies (filled below)
  p->h.tproj = h->tproj ? p->hTproj : nullptr;
  p->h.log_hetero_skedasticity = h->log_hetero_skedasticity ? p->hHet : nullptr;
  p->h.log_multiscales_m05 = h->log_multiscales_m05 ? p->hMs : nullptr;
  // the pinned block may still be the source of the previous evaluation's transfer (a pass 1 repeated without a finish)
  GPR_HIP(hipEventSynchronize(p->ev_hy));
  int64_t used = (int64_t)p->mp * p->d + 64;  // doubles of the block this evaluation uploads (a prefix)
  if (h->log_hetero_skedasticity) {  // Kernel.create: Vec.map exp, lib/cov_se_fat.ml:63-65
    for (int i = 0; i < p->m; ++i) p->hHet[i] = std::exp(h->log_hetero_skedasticity[i]);
    for (int i = p->m; i < p->mp; ++i) p->hHet[i] = 0.0;
    p->h.log_hetero_skedasticity = p->hHet;  // only used as a presence flag from here on
    used = (p->hHet - p->hy_host) + p->mp;
  }
  if (h->log_multiscales_m05) {
    // Kernel.create: exp v +. 0.5, lib/cov_se_fat.ml:66-69 ; Fortran d x m == [m][d]; padding rows 1
    for (int64_t i = 0; i < (int64_t)p->m * p->d; ++i) p->hMs[i] = std::exp(h->log_multiscales_m05[i]) + 0.5;
    for (int64_t i = (int64_t)p->m * p->d; i < (int64_t)p->mp * p->d; ++i) p->hMs[i] = 1.0;
    p->h.log_multiscales_m05 = p->hMs;  // presence flag from here on
    used = (p->hMs - p->hy_host) + (int64_t)p->mp * p->d;
  }
  CovParams& cp = p->cp;
  cp.kind = p->kind;
  cp.ms = h->log_multiscales_m05 ? p->ms : nullptr;
  cp.log_sf2 = h->log_sf2;
  cp.sf2 = std::exp(h->log_sf2);
  if (p->kind == GPRHIP_COV_SE_ISO) {
    cp.inv_ell2 = std::exp(-2.0 * h->log_ell);  // lib/cov_se_iso.ml:41-44
    cp.inv_ell2_05 = -0.5 * cp.inv_ell2;
  } else {
    cp.inv_ell2 = 1.0;
    cp.inv_ell2_05 = -0.5;
  }
  // inducing: Fortran d x m == point-major [m][d]; pad to mp rows with zeros
  std::memcpy(p->hZ, h->inducing, (size_t)p->m * p->d * sizeof(double));
  std::memset(p->hZ + (size_t)p->m * p->d, 0, (size_t)(p->mp - p->m) * p->d * sizeof(double));
  {  // centroid of the inducing points
    for (int k = 0; k < 64; ++k) p->hShift[k] = 0.0;
    for (int c = 0; c < p->m; ++c)
      for (int k = 0; k < p->d && k < 64; ++k) p->hShift[k] += p->hZ[(size_t)c * p->d + k];
    for (int k = 0; k < 64; ++k) p->hShift[k] /= p->m;
  }
  if (h->tproj) {
    std::memcpy(p->hTproj, h->tproj, (size_t)p->D * p->d * sizeof(double));  // borrowed pointer: copy before returning
    p->h.tproj = p->hTproj;
    used = std::max<int64_t>(used, (p->hTproj - p->hy_host) + (int64_t)p->D * p->d);
  }
  GPR_HIP(hipMemcpyAsync(p->hy_dev, p->hy_host, (size_t)used * sizeof(double), hipMemcpyHostToDevice, p->stream));
  GPR_HIP(hipEventRecord(p->ev_hy, p->stream));
  if (h->tproj) launch_project(p->X, p->n, p->D, p->d, p->tproj, p->P, p->stream);
}

// Split-K factor of the SYRK-shaped accumulations over training points.  Slices are dealt to the
// 8 XCDs (mfma_gemm.hip), so the factor is a multiple of 8.  Blocks of one slice share their operand
// rows through the XCD's L2 while they run in step.  Measured at n=1M, m=2048 (rocprofv3 FETCH_SIZE per launch,
// against 16.4 GB of V): 60 GB through the fabric for 2k-, 4k- and 8k-row slices alike once every workgroup of the
// launch runs the same loop (mfma_gemm.hip) -- the re-fetch factor is set by the stagger of workgroup start times
// against the L2 window, not by the slice length -- and the step is fastest with 8k..16k-row slices (381.6 ms;
// 385 ms at 4k and at 32k), so slices are kept near `slice_rows`.  Among nearby factors the one whose
// (tiles x slices / 8) fills whole residency rounds of an XCD (32 CUs x 2 blocks) is taken.
int pick_kslices(int mp, int64_t rows_p, int max_slices, int64_t slice_rows, bool f64) {
  const int nt = mp / TILE, tiles = nt * (nt + 1) / 2;
  const int slots = std::max(8, gemm_resident_slots() / 8);  // workgroups one XCD holds at a time (64 on MI355X)
  auto items_of = [&](int ks) {  // items of one XCD: off-diagonal tiles per slice + the diagonal tiles' own slices
    const int ksd = gemm_syrk_diag_slices(ks, f64, true);
    return (tiles - nt) * (ks / 8) + nt * ((ksd + 7) / 8);
  };
  if (rows_p < 32 * slice_rows) {
    // Mid-size shards (round 5): with few tiles (36 at m = 1024, 10 at m = 512) and few `slice_rows`-sized slices the
    // launch is one or two badly filled residency rounds -- 72 items per XCD for 64 slots at n = 100 000, m = 1024, ten
    // items at n = 50 000, m = 512.  Here the factor is chosen over the WHOLE feasible range by the time a launch of that
    // shape takes: residency rounds (the last, partial one at about half price while it leaves every CU a single
    // workgroup) x (fixed cost of an item + its rows at half a CU's matrix rate) + the slice sum that follows.
    const double t_fixed = 10.0, t_row = f64 ? 0.22 : 0.11;  // us per item; us per training point of a 128 x 128 tile item
    int best = 8;
    double best_t = 1e300;
    for (int ks = 8; ks <= max_slices / 8 * 8; ks += 8) {
      if ((int64_t)ks * TILE > rows_p && ks > 8) break;  // at least 128 training points per slice
      const int items = items_of(ks);
      const int full = items / slots, rem = items % slots;
      // a partly filled round behind full ones costs a whole round (the dispatcher refills CUs in pairs of slots as they fall
      // free, so its workgroups still share their CUs: profiles/r05_ts_nn_m1024.txt); a launch that never fills the chip
      // runs one workgroup per CU at nearly twice the speed
      const double rounds = full + (rem == 0 ? 0.0 : (full == 0 && rem <= slots / 2 ? 0.55 : 1.0));
      const double t_item = t_fixed + (double)rows_p / ks * t_row;
      const double t_sum = (double)ks * tiles * TILE * TILE * (f64 ? 8.0 : 4.0) / 3.0e6;  // us at 3 TB/s
      const double t = rounds * t_item + t_sum;
      if (t < best_t - 1e-9) {
        best_t = t;
        best = ks;
      }
    }
    return best;
  }
  int target = (int)((rows_p / slice_rows + 7) / 8 * 8);
  target = std::max(8, std::min(target, max_slices / 8 * 8));
  int best = target;
  double best_eff = 0.0;
  for (int ks = std::max(8, target - 16); ks <= std::min(max_slices / 8 * 8, target + 16); ks += 8) {
    if ((int64_t)ks * BK * 8 > rows_p && ks > 8) continue;
    const int items = items_of(ks);
    const double eff = (double)items / ((double)((items + slots - 1) / slots) * slots);
    if (eff > best_eff + 1e-9) {
      best_eff = eff;
      best = ks;
    }
  }
  return best;
}

// Training points per workgroup of the gradient kernels: grad_slab_rows (256 or 1024 by the accumulator rows), halved down
// to 64 while a launch -- min(n, chunk) rows -- would leave most of the chip idle.  GPRHIP_GRAD_SLAB overrides with one of
// the sizes the kernels and their partial buffers are exercised with: 64, 128, 256, 512, 1024 (anything else is rounded
// down to the next of these).
int pick_grad_slab(int col_rows, int64_t n, int64_t chunk, int mp) {
  int slab = grad_slab_rows(col_rows);
  const int64_t rows = std::min(n, chunk);
  while (slab > 64 && ((rows + slab - 1) / slab) * (int64_t)(mp / TILE) < 256) slab /= 2;
  if (const char* e = getenv("GPRHIP_GRAD_SLAB")) {
    const int want = atoi(e);
    slab = 64;
    while (slab < 1024 && slab * 2 <= want) slab *= 2;
  }
  return slab;
}

// Row chunking and split-K scratch of a problem: one place, used by the creation entry point and by the memory plan.
struct Sizing {
  int64_t chunk = 0, slice_rows = 8192;
  int nchunks = 0, kslices = 8;
};
Sizing problem_sizing(int64_t n, int mp, int64_t chunk_rows, bool f64) {
  Sizing z;
  int64_t chunk = chunk_rows > 0 ? chunk_rows : 131072;
  if (const char* e = getenv("GPRHIP_CHUNK_ROWS")) chunk = atoll(e);
  chunk = round_up(std::min<int64_t>(std::max<int64_t>(chunk, 1), round_up(n, TILE)), TILE);
  z.chunk = chunk;
  z.nchunks = (int)((n + chunk - 1) / chunk);
  // partial-sum buffers of the split-K SYRK launches: one m x m slice per `slice_rows` training points, capped at 40 GB
  // (fp32-bulk mode: fp32 accumulation runs over one slice before the fp64 slice sum.  Measured at config 3 against
  // the fp64 evaluation, tools/lab10.sh: 2048 / 4096 / 8192 / 16384-row slices give the evidence to 7.4e-8 / 7.3e-8 /
  // 5.6e-8 / 2.3e-8 and SYRK launches of 119.9 / 120.3 / 119.0 / 118.9 ms -- no reason for shorter slices than fp64's.)
  if (const char* e = getenv("GPRHIP_SLICE_ROWS")) z.slice_rows = std::max<int64_t>(1024, atoll(e));
  const int64_t mm8 = (int64_t)mp * mp * 8;
  z.kslices = (int)std::max<int64_t>(8, std::min<int64_t>((round_up(n, z.slice_rows) / z.slice_rows + 23) / 8 * 8, (40LL << 30) / mm8 / 8 * 8));
  if (n < 32 * z.slice_rows) {
    // mid-size shards take many short slices: exactly the factor pick_kslices will choose for this shape (it depends on
    // (m, rows, cap) only) within a cap of 128 slices / 2 GB -- not the cap itself, which at n = 100 000, m = 1024 was 1 GB
    // of scratch beside a 0.8 GB V store; the m x m products that borrow the scratch need eight slices
    const int cap = (int)std::max<int64_t>(8, std::min<int64_t>(128, (2LL << 30) / mm8 / 8 * 8));
    const int64_t rows_p = (int64_t)(z.nchunks - 1) * chunk + round_up(n - (int64_t)(z.nchunks - 1) * chunk, TILE);
    z.kslices = std::max(8, pick_kslices(mp, rows_p, cap, z.slice_rows, f64));
  }
  if (const char* e = getenv("GPRHIP_KSLICES")) z.kslices = std::max(8, atoi(e) / 8 * 8);
  return z;
}

// Bytes a problem holds on its device (gprhip_memory_plan).  Everything gprhip_problem_create allocates plus the V store
// the first evaluation adds; the optional copy of K_nm (Cov_se_fat with projection hypers) is listed but not counted -- it
// is only taken while it leaves room.
void memory_plan(int cov_kind, int precision, int64_t n, int D, int d, int m, int64_t chunk_rows, gprhip_memory_plan_t* o) {
  const int mp = (int)round_up(m, TILE);
  const bool f32 = precision == GPRHIP_F32_BULK, fat = cov_kind == GPRHIP_COV_SE_FAT;
  const int64_t esz = f32 ? 4 : 8, mm = (int64_t)mp * mp;
  const Sizing z = problem_sizing(n, mp, chunk_rows, !f32);
  const int64_t npad = (int64_t)z.nchunks * z.chunk;
  std::memset(o, 0, sizeof *o);
  o->chunk_rows = z.chunk;
  o->kslices = z.kslices;
  o->inputs = (n * D + npad + (fat ? n * d : 0)) * 8;
  o->row_vectors = npad * 8 * (5 + (fat ? 1 : 0)) + (f32 ? npad * 4 * 3 : 0);
  o->v_store = npad * mp * esz;
  o->k_store_optional = fat ? npad * mp * esz : 0;
  o->chunk_buffers = 2 * z.chunk * mp * esz + 2 * z.chunk * 4 * (mp / TILE) * 8;
  o->slices = (int64_t)z.kslices * mm * esz + (int64_t)z.kslices * mp * 8;
  o->mxm = 10 * mm * 8 + (f32 ? 3 * mm * 4 : 0) + (int64_t)mp * TILE * 8 + (int64_t)(mp / TILE) * TILE * TILE * 8;
  const int64_t col_rows = d + 1 + (fat ? D + d : 0), km_rows = d + 2 + (fat ? d : 0);
  const int gslab = pick_grad_slab((int)col_rows, n, z.chunk, mp);
  const int64_t nslab = (z.chunk + gslab - 1) / gslab;
  o->rest = (gprhip_exchange_len(cov_kind, D, d, m, 1) + gprhip_exchange_len(cov_kind, D, d, m, 2)) * 8 +
            nslab * col_rows * mp * 8 + (fat ? ((z.chunk + 255) / 256) * (int64_t)D * d * 8 : 0) +
            (int64_t)((m + km_slab_rows(m) - 1) / km_slab_rows(m)) * km_rows * mp * 8 + (int64_t)(8 + km_rows + col_rows) * mp * 8;
  o->total = o->inputs + o->row_vectors + o->v_store + o->chunk_buffers + o->slices + o->mxm + o->rest;
}

template <typename TS>
void cov_chunk(gprhip_problem* p, int c, TS* K, hipStream_t s = nullptr) {
  const int64_t rows = p->rows_of(c);
  const int64_t rows_p = round_up(rows, TILE);
  const double* pts = p->pts() + (int64_t)c * p->chunk * p->d;
  launch_cov_cross<TS>(p->cp, pts, (int)rows, (int)rows_p, p->Z, p->m, p->mp, p->d, K, s ? s : p->stream, p->zshift);
}

template <typename TS>
void do_pass1(gprhip_problem* p, const gprhip_hypers* h, int want_grad, int64_t n_total, double* ar1) {
  if (!h || !h->inducing) {
    set_error("gprhip: hypers/inducing pointer is NULL");
    throw HipFail{ST_BAD_ARG};
  }
  if (!p->have_inputs || (!p->have_targets && !h->model_only)) {
    set_error("gprhip: inputs/targets not set");
    throw HipFail{ST_STATE};
  }
  GPR_HIP(hipSetDevice(p->device));
  hipStream_t s = p->stream;
  // the state of the previous evaluation is void from here on; finish() re-validates it
  p->have_model = p->have_factors = false;
  p->stage = 0;
  p->x_last = nullptr;
  p->cond_km = -1.0;
  upload_hypers(p, h);
  p->want_grad = want_grad;
  p->n_total = n_total;
  const int mp = p->mp;
  const int64_t mm = (int64_t)mp * mp;
  double* ar1_c = ar1 + packed_upper_len(mp);
  double* ar1_tail = ar1_c + mp;
  if (!p->Vstore)  // V = K U^-1 for all rows of the shard stays resident (one SYRK launch; pass 2 re-reads it)
    p->Vstore = p->alloc<TS>((int64_t)p->nchunks * p->chunk * mp);
  TS* const Vstore = static_cast<TS*>(p->Vstore);
  TS* const bufA = static_cast<TS*>(p->bufA);
  TS* const slices = static_cast<TS*>(p->slices);
  // K resident (see Kstore): decided at the first gradient evaluation that can use it, kept for the problem's life
  const bool small = p->use_small();
  const bool mid = p->use_mid();
  if (small && !p->small_part) p->small_part = p->alloc<double>(small_part_len(p->d, p->D));
  if (mid && !p->mid_part) p->mid_part = p->alloc<double>(mid_part_len(mp, p->d, p->D));
  if (!p->Kstore && !p->kstore_tried && !small && !mid && want_grad && p->k_resident && p->kind == GPRHIP_COV_SE_FAT && h->tproj &&
      !h->log_multiscales_m05 && p->d <= 64 && p->D <= 64 && !p->grad_scalar) {
    p->kstore_tried = true;
    size_t free_b = 0, total_b = 0;
    const size_t need = (size_t)p->nchunks * p->chunk * mp * sizeof(TS);
    // taken only while it leaves 4 GB free AND stays below 40 % of the device's memory: the copy is held for the life of
    // the problem, and other problems (fitc_gp caches several per functor; other shards may share the device) must still
    // find room.  GPRHIP_K_RESIDENT=0 never keeps it.
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > need + (size_t(4) << 30) && need <= total_b / 5 * 2) {
      void* ptr = nullptr;
      if (hipMalloc(&ptr, need) == hipSuccess) {
        p->allocs.push_back(ptr);
        p->Kstore = ptr;
      } else {
        (void)hipGetLastError();  // no room after all: the gradient kernel recomputes K
      }
    }
  }
  TS* const Kstore = static_cast<TS*>(p->Kstore);

  const bool reuse = h->reuse_v != 0;
  if (reuse && !p->have_v) {
    set_error("gprhip: reuse_v set but this problem holds no V of a previous evaluation");
    throw HipFail{ST_STATE};
  }
  p->have_v = false;
  if (!reuse) p->have_k = p->small_k_valid = false;
  // K_nm of the first chunk does not depend on U: it is built on the second stream while the (latency-bound, few-CU)
  // factorisation and inversion of K_m run -- 0.4 ms of every evaluation, which is what a chunk's builder takes.
  // (Not under the per-stage timer, whose events sit on the main stream.)
  const bool cov0_ahead = !reuse && !p->timer.on && !small && !mid;
  if (cov0_ahead) {
    GPR_HIP(hipEventRecord(p->ev_fork, s));  // hypers, inducing points and projections are enqueued on s
    GPR_HIP(hipStreamWaitEvent(p->stream2, p->ev_fork, 0));
    cov_chunk<TS>(p, 0, Kstore ? Kstore : bufA, p->stream2);
    GPR_HIP(hipEventRecord(p->ev_join, p->stream2));
  }
  tstart(p, "km_chol");
  // the scalars and the two potrf flags behind them (single-block problems: every slot an evaluation reads is written by
  // the factorisation kernels themselves, flags included), and the accumulators of the exchange-1 tail (the small row pass
  // writes them outright)
  if (mp != TILE || p->engine_steps) GPR_HIP(hipMemsetAsync(p->scal, 0, (NSCAL + 2) * sizeof(double), s));
  // (the one-tile passes write c~ and the tail outright; with two tiles c~ still comes from the engine's launch, accumulated)
  if (!(small || (mid && mp == TILE)) || reuse) GPR_HIP(hipMemsetAsync(ar1_c, 0, (size_t)(mp + A1_TAIL) * sizeof(double), s));
  // K_m + (hetero) + jitter goes straight into the factor's buffer (kj is scratch of the finish stage only)
  // (tried for 65 .. 128 inducing points too, round 6: 16 384 entries on ONE CU cost 20 us at d = 8 where the cov_upper launch
  //  spread over the chip costs 4 + its 4 us gap -- K_m phase 45 -> 68 us; it stays a launch of its own above 64)
  if (mp == TILE && p->m <= 64 && p->d <= 16 && !p->engine_steps && !p->has_ms() && p->small_path) {
    PotrfKm g;  // few inducing points: the covariance is built inside the factorisation kernel (chol.hip, MODE 2)
    g.cp = p->cp; g.Z = p->Z; g.m = p->m; g.d = p->d; g.jitter = h->jitter;
    g.het = p->has_het() ? p->het : nullptr; g.km = p->km;
    launch_potrf_km(g, p->umat, p->uinv, p->info, s);
  } else {
    launch_cov_upper(p->cp, p->Z, p->m, mp, p->d, h->jitter, p->has_het() ? p->het : nullptr, p->km, p->umat, s);
    potrf_trtri(p, p->umat, p->uinv, p->wmat, p->info);  // U = chol(K_m + jitter), lib/fitc_gp.ml:53-57, and U^-1
  }
  if (p->f32) launch_to_float(p->uinv, p->uinv_f, mm, s);
  tstop(p);

  if constexpr (std::is_same<TS, double>::value) {
    if (small && !reuse) {
      // at most 64 inducing points: covariance, V, the row quantities and both accumulations in one kernel + one reduction
      // (small.hip)
      tstart(p, "p1_small");
      SmallPass1Args a;
      a.cp = p->cp; a.pts = p->pts(); a.Z = p->Z; a.uinv = p->uinv; a.y = h->model_only ? nullptr : p->y;
      a.rows = (int)p->n; a.rows_p = (int)round_up(p->n, TILE); a.m = p->m; a.mp = mp; a.d = p->d;
      a.sigma2 = h->sigma2;
      a.V = Vstore; a.r = p->r; a.is = p->is; a.yis = p->yis; a.part = p->small_part;
      if (!p->small_k && p->d <= 8 && !p->has_ms()) p->small_k = p->alloc<double>((int64_t)a.rows_p * 64);
      a.Kout = (p->d <= 8 && !p->has_ms()) ? p->small_k : nullptr;
      launch_small_pass1(a, ar1, ar1_c, ar1_tail, s);
      p->small_k_valid = a.Kout != nullptr;
      tstop(p);
      p->stage = 1;
      p->have_v = true;
      p->have_k = false;
      return;
    }
    if (mid && !reuse) {
      // one or two 128-column tiles of inducing points: the same in one kernel per row block with the triangular operand
      // streamed from memory (mid.hip); with two tiles the accumulations B~ and c~ stay with the engine's launch below
      tstart(p, "p1_mid");
      MidPass1Args a;
      a.cp = p->cp; a.pts = p->pts(); a.Z = p->Z; a.uinv = p->uinv; a.y = h->model_only ? nullptr : p->y;
      a.rows = (int)p->n; a.rows_p = (int)round_up(p->n, TILE); a.m = p->m; a.mp = mp; a.d = p->d;
      a.sigma2 = h->sigma2;
      a.V = Vstore; a.r = p->r; a.is = p->is; a.yis = p->yis; a.part = p->mid_part;
      launch_mid_pass1(a, ar1, ar1_c, ar1_tail, s);
      tstop(p);
      if (mp > TILE && p->use_mid_gram()) {
        // two tiles: B~_part and c~ over the resident V by mid.hip's Gram launch pair (the engine's SYRK-shaped launch, its
        // column-sum reduction and its slice sum cost 45 us at these sizes whatever the row count)
        tstart(p, "p1_syrk_B");
        if (!p->mid_gram_part) p->mid_gram_part = p->alloc<double>(mid_gram_part_len());
        MidGramArgs ga;
        ga.V = reinterpret_cast<const double*>(Vstore); ga.w = p->is; ga.y = p->yis; ga.rows = (int)p->n;
        ga.part = p->mid_gram_part;
        if (p->timer.kernel) {
          if (!p->timer.k0) {
            GPR_HIP(hipEventCreate(&p->timer.k0));
            GPR_HIP(hipEventCreate(&p->timer.k1));
          }
          GPR_HIP(hipEventRecord(p->timer.k0, s));
        }
        launch_mid_gram(ga, ar1, ar1_c, s);
        if (p->timer.kernel) {
          GPR_HIP(hipEventRecord(p->timer.k1, s));
          p->timer.k_recorded = true;
        }
        tstop(p);
      }
      if (mp == TILE || p->use_mid_gram()) {
        p->stage = 1;
        p->have_v = true;
        p->have_k = false;
        return;
      }
    }
  }
  const bool rows_done = mid && !reuse;  // (two tiles: V and the row quantities are there, the chunk loop has nothing to do)
  // Overlap (GPRHIP_COV_OVERLAP=1): chunk c + 1's covariance goes out on the second stream just before chunk c's V product
  // goes out on the main one, into the other chunk buffer (pass 1 uses one of the two at a time) -- it waits only for the V
  // product that last read that buffer, i.e. it runs beside V(c).
#ifdef GPRHIP_LAB
  const bool overlap = p->cov_overlap && cov0_ahead && p->nchunks >= 2;
#else
  constexpr bool overlap = false;  // (measured: no gain, profiles/r05_lab_cov_overlap.txt -- lab build only)
#endif
  TS* const bufB1 = static_cast<TS*>(p->bufB);
  auto kbuf = [&](int c) -> TS* {
    if (Kstore) return Kstore + (int64_t)c * p->chunk * mp;
    return (overlap && (c & 1)) ? bufB1 : bufA;
  };
  for (int c = 0; c < (rows_done ? 0 : p->nchunks); ++c) {
    const int64_t rows = p->rows_of(c);
    const int rows_p = (int)round_up(rows, TILE);
    const int64_t base = (int64_t)c * p->chunk;
    TS* V = Vstore + base * mp;
    TS* const Kc = kbuf(c);  // this chunk's K_nm: kept, or in a chunk buffer
    if (!reuse) {
      tstart(p, "p1_cov");
      if (overlap) {
        if (c + 1 < p->nchunks) {
          if (c >= 1 && !Kstore) GPR_HIP(hipStreamWaitEvent(p->stream2, p->ev_vdone[(c - 1) & 1], 0));
          cov_chunk<TS>(p, c + 1, kbuf(c + 1), p->stream2);
          GPR_HIP(hipEventRecord(p->ev_cov[(c + 1) & 1], p->stream2));
        }
        GPR_HIP(hipStreamWaitEvent(s, c == 0 ? p->ev_join : p->ev_cov[c & 1], 0));
      } else if (c == 0 && cov0_ahead) GPR_HIP(hipStreamWaitEvent(s, p->ev_join, 0));
      else cov_chunk<TS>(p, c, Kc);
      tstop(p);
      tstart(p, "p1_trmm_V");
      GemmArgsT<TS> g;  // V = K U^-1   (dtrsm `R, lib/fitc_gp.ml:226-227)
      g.A = Kc; g.lda = mp; g.B = inv_u<TS>(p); g.ldb = mp; g.C = V; g.ldc = mp;
      g.M = rows_p; g.N = mp; g.K = mp; g.tri = TRI_KHI_BN; g.order = p->tile_order;
      g.rp_sumsq = p->rp1;  // r = k_diag - rowsum(V.^2) comes out of the epilogue (Mat.syrk_diag, :222-223)
      launch_gemm(OP_NN, g, s);
      if (overlap) GPR_HIP(hipEventRecord(p->ev_vdone[c & 1], s));
      tstop(p);
    }
    tstart(p, "p1_rows");
    Pass1RowArgs ra;
    ra.part = reuse ? nullptr : p->rp1; ra.npart = gemm_row_parts_per_tile() * (mp / TILE); ra.ld = rows_p;
    ra.y = h->model_only ? nullptr : p->y + base; ra.rows = (int)rows;
    ra.sf2 = p->cp.sf2; ra.sigma2 = h->sigma2;
    ra.r = p->r + base; ra.is = p->is + base; ra.yis = p->yis + base; ra.partial = p->rowpart;
    launch_pass1_rows(ra, s);
    launch_reduce_rows(p->rowpart, pass1_row_blocks(rows_p), 4, ar1_tail, 1, s);
    tstop(p);
  }
  if (mid && reuse && mp > TILE && p->use_mid_gram()) {
    // Model.update_sigma2 on a two-tile problem whose fresh evaluations accumulate B~ with mid.hip's launch pair: the same
    // pair here, so that the re-weighted evaluation returns the very numbers a fresh one would
    // (test_update_sigma2_reuses_resident_v)
    tstart(p, "p1_syrk_B");
    if (!p->mid_gram_part) p->mid_gram_part = p->alloc<double>(mid_gram_part_len());
    MidGramArgs ga;
    ga.V = reinterpret_cast<const double*>(Vstore); ga.w = p->is; ga.y = p->yis; ga.rows = (int)p->n;
    ga.part = p->mid_gram_part;
    launch_mid_gram(ga, ar1, ar1_c, s);
    tstop(p);
    p->stage = 1;
    p->have_v = true;
    return;
  }
  // one SYRK-shaped launch over all rows of the shard (V is resident): B~_part = V^T diag(is) V
  // (R~^T R~ replaces the stacked QR's R, lib/fitc_gp.ml:170-182), and c~ = V^T (is .* y)
  const int64_t ktot = p->rows_total_padded();
  const int ks = pick_kslices(mp, ktot, p->kslices, p->slice_rows, !p->f32);
  tstart(p, "p1_syrk_B");
  GemmArgsT<TS> b;
  b.A = Vstore; b.lda = mp; b.B = Vstore; b.ldb = mp; b.C = slices; b.ldc = mp;
  b.M = mp; b.N = mp; b.K = (int)ktot; b.beta = 0.0; b.scale_k = row_weights<TS>(p, p->is, p->is_f); b.upper_only = 1;
  b.kslices = ks; b.slice_stride = mm;
  // c~ rides along: the diagonal-tile blocks of every k-slice also sum V[k][c] * (is*y)[k] over their rows
  b.cs_w = row_weights<TS>(p, p->yis, p->yis_f); b.cs_out = p->gemvpart;
  if (p->timer.kernel) {
    if (!p->timer.k0) {
      GPR_HIP(hipEventCreate(&p->timer.k0));
      GPR_HIP(hipEventCreate(&p->timer.k1));
    }
    GPR_HIP(hipEventRecord(p->timer.k0, s));
  }
  launch_gemm(OP_TN, b, s);
  if (p->timer.kernel) {
    GPR_HIP(hipEventRecord(p->timer.k1, s));
    p->timer.k_recorded = true;
  }
  const int ksd = gemm_syrk_diag_slices(ks, !p->f32, true);  // the diagonal tiles (which also store the column sums) use fewer, longer slices
  launch_reduce_rows(p->gemvpart, ksd, mp, ar1_c, 1, s);
  tstop(p);
  p->ks_used = ks;
  launch_sum_slices<TS>(nullptr, slices, ks, mm, mp, ar1, s, 1, ksd);
  p->stage = 1;
  p->have_v = true;  // (revoked by finish() if the factorisation of K_m turns out to have failed)
  if (!reuse) p->have_k = Kstore != nullptr && !rows_done;
}

template <typename TS>
void do_pass2(gprhip_problem* p, const double* ar1, double* ar2) {
  if (p->stage != 1) {
    set_error("gprhip: eval_pass2 called before eval_pass1");
    throw HipFail{ST_STATE};
  }
  GPR_HIP(hipSetDevice(p->device));
  hipStream_t s = p->stream;
  const int mp = p->mp;
  const int64_t mm = (int64_t)mp * mp;
  const double* ar1_c = ar1 + packed_upper_len(mp);
  double* ar2_col = ar2 + packed_upper_len(mp);
  double* ar2_proj = ar2_col + p->col_rows() * mp;
  double* ar2_tail = ar2_proj + (int64_t)p->dbig() * p->d;
  const bool mo = p->h.model_only != 0;
  const bool proj = p->has_proj();

  tstart(p, "b_chol");
  // B~ = I + sum of shard parts; R~ = chol(B~): R = R~ U is the reference's r_mat (lib/fitc_gp.ml:181)
  const bool fused_b = mp == TILE && !p->engine_steps;
  p->a1_in_scal = fused_b;
  if (fused_b) {
    // a single block: I + the accumulation, the factorisation, the inverse and the m-vectors below in one kernel
    PotrfFuse f;
    f.src = ar1; f.cvec = ar1_c;
    f.tail_in = ar1_c + mp; f.tail_out = p->scal + SC_A1TAIL;
    f.uinv = p->uinv; f.bvec = p->bvec; f.ttil = p->ttil; f.tvec = p->tvec;
    f.logdet = p->scal + SC_LOGDET_B; f.bb = p->scal + SC_BB;
    launch_potrf_fused(f, p->bmat, p->rinv, p->info + 1, p->m, s);
  } else {
    hipLaunchKernelGGL(add_identity_upper_kernel, dim3((mp + 255) / 256, mp), dim3(256), 0, s, ar1, mp,
                       p->bmat);
    potrf_trtri(p, p->bmat, p->rinv, p->wmat, p->info + 1);
  }
  if (p->f32) launch_to_float(p->rinv, p->rinv_f, mm, s);
  TS* const Vstore = static_cast<TS*>(p->Vstore);
  TS* const bufA = static_cast<TS*>(p->bufA);
  TS* const bufB = static_cast<TS*>(p->bufB);
  TS* const slices = static_cast<TS*>(p->slices);
  // b = R~^-T c~ (= Q_n^T y~, lib/fitc_gp.ml:285-286);  t~ = R~^-1 b;  t = U^-1 t~ (trsv, :291 / :1167)
  if (!fused_b) {
    // (log|B~| and |b|^2 ride on the first two launches: two launches less on the chain)
    launch_triu_matvec_rider(p->rinv, mp, ar1_c, p->bvec, 1, 1, p->bmat, mp, p->scal + SC_LOGDET_B, s);
    launch_triu_matvec_rider(p->rinv, mp, p->bvec, p->ttil, 0, 2, p->bvec, mp, p->scal + SC_BB, s);
    launch_triu_matvec(p->uinv, mp, p->ttil, p->tvec, 0, s);
  }
  tstop(p);

  // evidence-only evaluations (multim_f) carry nothing in the second exchange buffer: only its scalar tail is cleared,
  // and the caller need not reduce it
  const bool small = p->use_small();  // its reduction writes every entry of the exchange-2 buffer
  const bool mid = p->use_mid();      // (the same)
  // (two tiles through mid.hip: its reduction writes everything behind the packed tiles, the engine's slice sum the tiles)
  if (p->want_grad && !small && !mid) GPR_HIP(hipMemsetAsync(ar2, 0, (size_t)gprhip_ar2_len(p) * sizeof(double), s));
  else if (!p->want_grad) GPR_HIP(hipMemsetAsync(ar2_tail, 0, (size_t)A2_TAIL * sizeof(double), s));
  if constexpr (std::is_same<TS, double>::value) {
    if (p->want_grad && small) {
      // at most 64 inducing points: Q', the row quantities, X~, X, the column sums of E = X .* K and G~ in one kernel (small.hip);
      // B~^-1 is formed by the finish kernel, R^-1 is not needed
      tstart(p, "p2_small");
      if (!p->small_part) p->small_part = p->alloc<double>(small_part_len(p->d, p->D));
      p->merged_x = false;
      SmallPass2Args a;
      a.cp = p->cp; a.pts = p->pts(); a.Z = p->Z; a.uinv = p->uinv; a.rinv = p->rinv; a.bvec = p->bvec; a.ttil = p->ttil;
      a.V = Vstore; a.y = mo ? nullptr : p->y; a.is = p->is; a.r = p->r;
      a.Kin = (p->small_k_valid && p->d <= 8 && !p->has_ms()) ? p->small_k : nullptr;
      a.big = proj ? p->X : nullptr; a.D = proj ? p->D : 0;
      a.rows = (int)p->n; a.rows_p = (int)round_up(p->n, TILE); a.m = p->m; a.mp = mp; a.d = p->d;
      a.variational = p->h.variational;
      // (X itself is only kept for the debug fetch "x_rows", which wants all rows in one chunk buffer)
      a.w = p->w; a.v = p->v; a.es = proj ? p->es : nullptr; a.X = p->nchunks == 1 ? bufB : nullptr; a.part = p->small_part;
      launch_small_pass2(a, (int)p->col_rows(), ar2, ar2_col, ar2_proj, ar2_tail, s);
      p->x_last = a.X;
      tstop(p);
      p->stage = 2;
      return;
    }
    if (p->want_grad && mid) {
      // one or two 128-column tiles: Q', the row quantities, X~, X, E = X .* K with its moments (and, one tile, G~) in one
      // kernel (mid.hip); B~^-1 is formed by the finish kernels, R^-1 is not needed
      tstart(p, "p2_mid");
      if (!p->mid_part) p->mid_part = p->alloc<double>(mid_part_len(mp, p->d, p->D));
      p->merged_x = false;
      // (U^-T and R~^-T for the "times B^T" products and the finish stage: W~'s and B~^-1's buffers are free on this path)
      launch_mid_transposes(p->uinv, p->rinv, mp, p->wtil, p->binv, s);
      MidPass2Args a;
      a.cp = p->cp; a.pts = p->pts(); a.Z = p->Z; a.rinv = p->rinv; a.uinvT = p->wtil; a.rinvT = p->binv;
      a.bvec = p->bvec; a.ttil = p->ttil;
      a.V = Vstore; a.y = mo ? nullptr : p->y; a.is = p->is; a.r = p->r;
      a.big = proj ? p->X : nullptr; a.D = proj ? p->D : 0; a.shift = p->zshift;
      a.rows = (int)p->n; a.rows_p = (int)round_up(p->n, TILE); a.m = p->m; a.mp = mp; a.d = p->d;
      a.variational = p->h.variational;
      a.w = p->w; a.v = p->v; a.es = proj ? p->es : nullptr; a.X = p->nchunks == 1 ? bufB : nullptr; a.part = p->mid_part;
      launch_mid_pass2(a, (int)p->col_rows(), ar2, ar2_col, ar2_proj, ar2_tail, s);
      p->x_last = a.X;
      if (proj) {  // second term of the `Proj derivative from the per-row sums of E the kernel left (as the engine path)
        for (int c = 0; c < p->nchunks; ++c) {
          const int64_t rows = p->rows_of(c), base = (int64_t)c * p->chunk;
          launch_proj_term2(p->X + base * p->D, p->P + base * p->d, p->es + base, 1, (int)rows, p->D, p->d, p->projpart, s);
          launch_reduce_rows(p->projpart, (int)((rows + 255) / 256), p->D * p->d, ar2_proj, 1, s);
        }
      }
      tstop(p);
      if (mp > TILE && p->use_mid_gram()) {  // two tiles: G~_part = V^T diag(v) V by the Gram launch pair of mid.hip
        tstart(p, "p2_syrk_W");
        if (!p->mid_gram_part) p->mid_gram_part = p->alloc<double>(mid_gram_part_len());
        MidGramArgs ga;
        ga.V = reinterpret_cast<const double*>(Vstore); ga.w = p->v; ga.y = nullptr; ga.rows = (int)p->n;
        ga.part = p->mid_gram_part;
        launch_mid_gram(ga, ar2, nullptr, s);
        tstop(p);
      } else if (mp > TILE) {  // ... or from the engine, as below (GPRHIP_MID_GRAM=0)
        tstart(p, "p2_syrk_W");
        GemmArgsT<TS> wg;
        wg.A = Vstore; wg.lda = mp; wg.B = Vstore; wg.ldb = mp; wg.C = slices; wg.ldc = mp;
        wg.M = mp; wg.N = mp; wg.K = (int)p->rows_total_padded(); wg.beta = 0.0;
        wg.scale_k = row_weights<TS>(p, p->v, p->v_f); wg.upper_only = 1;
        wg.kslices = p->ks_used; wg.slice_stride = mm;
        const bool as_ws = p->w_as_ws != 0;
        if (as_ws) {
          wg.cs_w = wg.scale_k;
          wg.cs_out = p->gemvpart;
        }
        launch_gemm(OP_TN, wg, s);
        tstop(p);
        launch_sum_slices<TS>(nullptr, slices, p->ks_used, mm, mp, ar2, s, 1, gemm_syrk_diag_slices(p->ks_used, !p->f32, as_ws));
      }
      p->stage = 2;
      return;
    }
  }
  if (p->want_grad) {
    // B~^-1 = R~^-1 R~^-T (needed by the finish stage only) and R^-1 = U^-1 R~^-1 (needed by the first X product) do not
    // belong on the chain between the factorisation and the Q' products: they go to the second stream and run beside
    // the first chunk's Q' launch (0.28 ms of every gradient evaluation at m = 2048).  Both use the split-K scratch,
    // which the main stream touches next in the pass-2 SYRK -- it waits for ev_binv before that.  (Under the per-stage
    // timer everything stays on the main stream, so that "inverses" keeps its meaning.)
    const bool side = !p->timer.on;
    hipStream_t si = side ? p->stream2 : s;
    if (side) {
      GPR_HIP(hipEventRecord(p->ev_fork, s));  // R~^-1 (and its fp32 copy), t~, t are enqueued on s
      GPR_HIP(hipStreamWaitEvent(p->stream2, p->ev_fork, 0));
    }
    tstart(p, "inverses");
    // The two-phase X product (below) needs R^-1 = U^-1 R~^-1, one more m x m product (0.14 ms at m = 2048, 0.8 ms at
    // m = 4096), and saves 2.5 us per 1000 training points at m = 2048 (6 us at m = 4096): taken from 48 m training
    // points per shard on.
    p->merged_x = p->merged_x_mode == 2 || (p->merged_x_mode == 1 && p->n >= 48 * (int64_t)p->m);
    if (p->merged_x) {
      GemmArgs rf;  // R^-1 = U^-1 R~^-1, both upper triangular
      rf.A = p->uinv; rf.lda = mp; rf.B = p->rinv; rf.ldb = mp; rf.C = p->rfinv; rf.ldc = mp;
      rf.M = mp; rf.N = mp; rf.K = mp; rf.tri = TRI_BAND; rf.upper_only = 1;
      // the corner tile's k-range is the whole of m: eight k-slices keep the launch from waiting on it
      const int rks = (mp >= 1024 && 8 * mm * 8 <= p->slices_bytes) ? 8 : 1;
      if (rks > 1) {
        rf.C = static_cast<double*>(p->slices);
        rf.kslices = rks;
        rf.slice_stride = mm;
        launch_gemm(OP_NN, rf, si);
        launch_sum_slices<double>(nullptr, static_cast<double*>(p->slices), rks, mm, mp, p->rfinv, si);
      } else {
        launch_gemm(OP_NN, rf, si);
      }
      if (p->f32) launch_to_float(p->rfinv, p->rfinv_f, mm, si);
    }
    if (side) GPR_HIP(hipEventRecord(p->ev_rf, si));
    triu_xxt(p, p->rinv, p->binv, si);  // B~^-1 (upper tiles)
    if (side) GPR_HIP(hipEventRecord(p->ev_binv, si));
    tstop(p);
    bool derive_inducing = false;
    for (int c = 0; c < p->nchunks; ++c) {
      const int64_t rows = p->rows_of(c);
      const int rows_p = (int)round_up(rows, TILE);
      const int64_t base = (int64_t)c * p->chunk;
      const TS* V = Vstore + base * mp;
      tstart(p, "p2_trmm_Q");
      GemmArgsT<TS> q;  // Q' = V R~^-1 = K R^-1  (Q_n = diag(sqrt is) Q', lib/fitc_gp.ml:176-182)
      q.A = V; q.lda = mp; q.B = inv_r<TS>(p); q.ldb = mp; q.C = bufA; q.ldc = mp;
      q.M = rows_p; q.N = mp; q.K = mp; q.tri = TRI_KHI_BN; q.order = p->tile_order;
      q.rp_sumsq = p->rp1; q.rp_dot = p->rp2; q.rp_vec = p->bvec;  // q_diag and Q'b from the epilogue
      launch_gemm(OP_NN, q, s);
      tstop(p);
      tstart(p, "p2_rows");
      Pass2RowArgs ra;
      ra.part_sq = p->rp1; ra.part_dot = p->rp2; ra.npart = gemm_row_parts_per_tile() * (mp / TILE); ra.ld = rows_p;
      ra.y = mo ? nullptr : p->y + base; ra.is = p->is + base; ra.r = p->r + base;
      ra.rows = (int)rows; ra.variational = p->h.variational;
      ra.sf2 = p->cp.sf2; ra.es = proj ? p->es + base : nullptr;
      ra.w = p->w + base; ra.v = p->v + base; ra.partial = p->rowpart;
      launch_pass2_rows(ra, s);
      launch_reduce_rows(p->rowpart, pass1_row_blocks(rows_p), 4, ar2_tail, 1, s);
      tstop(p);
      const TS* Xc;  // X of this chunk
      if (p->merged_x) {
        // X = diag(is) Q' R^-T - diag(v) V U^-T - w t^T  (S, U_mat and the ger of lib/fitc_gp.ml:931-939, :1204-1206) as
        // one launch of two-phase items: acc = V U^-T, rows scaled by -v/is, acc += Q' R^-T, epilogue is*acc - w t^T --
        // one epilogue per tile instead of two, no X~ round trip, no operand read in the epilogue
        if (side && c == 0) GPR_HIP(hipStreamWaitEvent(s, p->ev_rf, 0));
        tstart(p, "p2_trmm_SX");
        GemmArgsT<TS> xg;
        xg.A = bufA; xg.lda = mp; xg.B = inv_rfull<TS>(p); xg.ldb = mp; xg.C = bufB; xg.ldc = mp;
        xg.A2 = V; xg.B2 = inv_u<TS>(p); xg.mid_num = p->v + base; xg.mid_den = p->is + base;
        xg.M = rows_p; xg.N = mp; xg.K = mp; xg.tri = TRI_KLO_BN; xg.order = p->tile_order;
        xg.epi_rows_a = p->is + base; xg.epi_rows_c = p->w + base; xg.epi_col = p->tvec;
        launch_gemm(OP_NT, xg, s);
        tstop(p);
        Xc = bufB;
      } else {
        tstart(p, "p2_trmm_S");
        GemmArgsT<TS> sg;  // X~ = diag(is) Q' R~^-T - diag(v) V - w t~^T   (S, U_mat and the ger of :936-938, :1204-1206)
        sg.A = bufA; sg.lda = mp; sg.B = inv_r<TS>(p); sg.ldb = mp; sg.C = bufB; sg.ldc = mp;
        sg.M = rows_p; sg.N = mp; sg.K = mp; sg.tri = TRI_KLO_BN; sg.order = p->tile_order;
        sg.epi_rows_a = p->is + base; sg.epi_rows_b = p->v + base; sg.epi_rows_c = p->w + base;
        sg.epi_col = p->ttil; sg.epi_mat = V; sg.epi_ldm = mp;
        launch_gemm(OP_NT, sg, s);
        tstop(p);
        tstart(p, "p2_trmm_X");
        GemmArgsT<TS> xg;  // X = X~ U^-T
        xg.A = bufB; xg.lda = mp; xg.B = inv_u<TS>(p); xg.ldb = mp; xg.C = bufA; xg.ldc = mp;
        xg.M = rows_p; xg.N = mp; xg.K = mp; xg.tri = TRI_KLO_BN; xg.order = p->tile_order;
        launch_gemm(OP_NT, xg, s);
        tstop(p);
        Xc = bufA;
      }
      p->x_last = (p->nchunks == 1) ? static_cast<const void*>(Xc) : nullptr;
      tstart(p, "p2_grad");
      GradArgs<TS> ga;
      ga.X = Xc; ga.pts = p->pts() + base * p->d; ga.Z = p->Z;
      ga.rows = (int)rows; ga.rows_p = rows_p; ga.m = p->m; ga.mp = mp; ga.d = p->d;
      ga.log_sf2 = p->cp.log_sf2; ga.inv_ell2_05 = p->cp.inv_ell2_05;
      ga.colpart = p->colpart; ga.scalpart = p->scalpart;
      ga.big = proj ? p->X + base * p->D : nullptr; ga.D = proj ? p->D : 0;
      ga.ms = p->cp.ms; ga.rowes = nullptr; ga.shift = p->zshift;
      ga.K = (p->Kstore && p->have_k && proj && !ga.ms) ? static_cast<const TS*>(p->Kstore) + base * mp : nullptr;
      ga.col_rows = p->d + 1 + ga.D + (ga.ms ? p->d : 0);  // rows this launch produces (tightly packed)
      ga.slab = p->grad_slab;
      const int nslots = 4 * ((mp + 255) / 256);
      if (ga.ms && proj) {
        if (!p->rowes) {
          p->rowes = p->alloc<double>(p->chunk * nslots * p->d);
          p->es2 = p->alloc<double>(p->chunk * p->d);
        }
        ga.rowes = p->rowes;
      }
      // matrix-core version unless multiscales (or > 64 dimensions) need the scalar kernel; GPRHIP_GRAD_SCALAR=1
      // forces the scalar one (parity tests run both)
      int nbx = p->grad_scalar ? 0 : grad_mfma_col_blocks(ga);
      if (p->d > 64 || ga.D > 64) {
        // wide points: K of the chunk is rebuilt into the chunk buffer that does not hold X, and E = X .* K read from memory
        TS* const Kw = (Xc == bufA) ? bufB : bufA;
        cov_chunk<TS>(p, c, Kw);
        launch_grad_wide(ga, static_cast<const TS*>(Kw), s);
        nbx = (mp + 255) / 256;
      } else if (nbx > 0) {
        launch_grad_mfma(ga, s);
        derive_inducing = proj;  // that kernel accumulates only the projection-gradient sums (grad_mfma.hip)
      } else {
        launch_grad_fused(ga, s);
        nbx = (mp + 255) / 256;
      }
      const int nslabs = (int)((rows + ga.slab - 1) / ga.slab);
      launch_reduce_rows(p->colpart, nslabs, ga.col_rows * mp, ar2_col, 1, s);
      if (proj) {
        if (ga.ms) {
          launch_reduce_rowes(p->rowes, (int)rows, nslots, p->d, p->es2, s);
          launch_proj_term2(p->X + base * p->D, p->P + base * p->d, p->es2, p->d, (int)rows, p->D, p->d,
                            p->projpart, s);
        } else {
          launch_proj_term2(p->X + base * p->D, p->P + base * p->d, p->es + base, 1, (int)rows, p->D, p->d,
                            p->projpart, s);
        }
        launch_reduce_rows(p->projpart, (int)((rows + 255) / 256), p->D * p->d, ar2_proj, 1, s);
      }
      launch_reduce_rows(p->scalpart, nslabs * nbx, 2, ar2_tail + A2_SUME, 1, s);
      tstop(p);
    }
    if (derive_inducing) launch_proj_inducing_grad(ar2_col, mp, p->d, p->D, p->tproj, s);
    // G~_part = V^T diag(v) V over all rows (the two dsyrk of lib/fitc_gp.ml:1198-1203, whitened, in one)
    const int64_t ktot = p->rows_total_padded();
    if (side) GPR_HIP(hipStreamWaitEvent(s, p->ev_binv, 0));  // B~^-1 done: the split-K scratch is free again
    tstart(p, "p2_syrk_W");
    GemmArgsT<TS> wg;
    wg.A = Vstore; wg.lda = mp; wg.B = Vstore; wg.ldb = mp; wg.C = slices; wg.ldc = mp;
    wg.M = mp; wg.N = mp; wg.K = (int)ktot; wg.beta = 0.0; wg.scale_k = row_weights<TS>(p, p->v, p->v_f); wg.upper_only = 1;
    wg.kslices = p->ks_used; wg.slice_stride = mm;
    // The pass-2 launch goes through the kernel of the pass-1 one (its diagonal tiles then also form column sums nobody
    // reads): measured in round 4, the plain weighted kernel (gemm_*_tn_w) runs this very launch with 172 GB instead of
    // 67 GB through the fabric and 1-2 ms slower on most boxes of the pool -- its workgroups fall out of step from the
    // first residency round on -- while the column-sum kernel does not (tools/lab19.sh, lab20.sh; GPRHIP_W_AS_WS=0 restores
    // the plain kernel).
    const bool as_ws = p->w_as_ws != 0;
    if (as_ws) {
      wg.cs_w = wg.scale_k;
      wg.cs_out = p->gemvpart;
    }
    launch_gemm(OP_TN, wg, s);
    tstop(p);
    launch_sum_slices<TS>(nullptr, slices, p->ks_used, mm, mp, ar2, s, 1, gemm_syrk_diag_slices(p->ks_used, !p->f32, as_ws));
  }
  p->stage = 2;
}

// Finish stage, first half: the m x m work of the gradient (W, traces) and the asynchronous copies of everything the
// host assembly needs, all on the problem's stream; nothing blocks.  `light` (shards of a multi-device context other
// than the first): only the factorisation flags are fetched -- the reduced buffers are identical on every device, and
// one device's assembly serves the caller.
void do_finish_enqueue(gprhip_problem* p, const double* ar2, bool light = false) {
  if (p->stage != 2) {
    set_error("gprhip: eval_finish called before eval_pass2");
    throw HipFail{ST_STATE};
  }
  GPR_HIP(hipSetDevice(p->device));
  hipStream_t s = p->stream;
  const int mp = p->mp, m = p->m, d = p->d;
  const double* ar2_col = ar2 + packed_upper_len(mp);
  const double* ar2_proj = ar2_col + p->col_rows() * mp;
  const double* ar2_tail = ar2_proj + (int64_t)p->dbig() * d;
  const int nkslab = (m + km_slab_rows(m) - 1) / km_slab_rows(m);
  p->stage = 3;
  if (light) {
    GPR_HIP(hipMemcpyAsync(p->res_host + NSCAL, p->info, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
    return;
  }
  const bool wdiag = p->want_grad && (p->has_het() || p->has_ms());
  const int64_t n_a2 = (ar2_tail + A2_TAIL) - ar2_col;  // the exchange-2 buffer from its column block on
  if (p->want_grad && p->use_small()) {
    // at most 64 inducing points: the m x m work in one workgroup, which also gathers the exchange-2 tail behind the
    // result block
    tstart(p, "finish");
    SmallFinishArgs a;
    a.uinv = p->uinv; a.rinv = p->rinv; a.ttil = p->ttil; a.km = p->km; a.Z = p->Z; a.g = ar2;
    a.ms = p->has_ms() ? p->ms : nullptr;
    a.m = m; a.mp = mp; a.d = d; a.km_rows = p->has_ms() ? 2 * d + 2 : d + 2;
    a.wmat = p->wmat; a.kmred = p->kmred; a.wdiag = wdiag ? p->wdiag : nullptr;
    a.gather_from = ar2_col; a.n_gather = n_a2; a.ex = p->ex_dev + A1_TAIL;
    launch_small_finish(a, s);
    tstop(p);
    const int64_t n_res = p->res_len + A1_TAIL + n_a2;
    if (n_res * (int64_t)sizeof(double) > 32768) {  // (d > 8 or so: above 32 KB a copy starts 17 us late, a kernel at once)
      ShipArgs sh;
      sh.src[0] = p->res_dev; sh.dst[0] = p->res_host; sh.n[0] = n_res;
      launch_ship(sh, s);
    } else {
      GPR_HIP(hipMemcpyAsync(p->res_host, p->res_dev, (size_t)n_res * sizeof(double), hipMemcpyDeviceToHost, s));
    }
    return;
  }
  if (p->want_grad && p->use_mid()) {
    // one or two 128-blocks: the m x m work in two launches of one workgroup per 16 rows (mid.hip), the first of which also
    // gathers the exchange-2 tail behind the result block
    tstart(p, "finish");
    MidFinishArgs a;
    a.uinv = p->uinv; a.rinv = p->rinv; a.uinvT = p->wtil; a.rinvT = p->binv; a.ttil = p->ttil; a.km = p->km; a.Z = p->Z;
    a.g = ar2;
    a.m = m; a.mp = mp; a.d = d; a.km_rows = d + 2;
    a.wmat = p->wmat; a.kmred = p->kmred; a.wdiag = wdiag ? p->wdiag : nullptr;
    a.ybuf = p->kj;  // (free after the factorisation of K_m)
    a.gather_from = ar2_col; a.n_gather = n_a2; a.ex = p->ex_dev + A1_TAIL;
    // the kernels' last workgroup writes the result block (and, two tiles, the exchange-1 tail) into the pinned mirror itself
    if (!p->mid_done) {
      p->mid_done = p->alloc<int>(1);
      GPR_HIP(hipMemsetAsync(p->mid_done, 0, sizeof(int), s));
    }
    a.res_dev = p->res_dev; a.res_host = p->res_host; a.res_total = p->res_len + A1_TAIL + n_a2; a.done_ctr = p->mid_done;
    if (!p->a1_in_scal) {  // (two tiles: the B~ phase is not the fused single-block kernel that leaves the tail in the result block)
      a.a1_tail = p->ar1 + packed_upper_len(mp) + mp;
      a.a1_host = p->ex_host;
    }
    launch_mid_finish(a, s);
    tstop(p);
    return;
  }
  if (p->want_grad) {
    tstart(p, "finish");
    launch_build_w(p->binv, p->ttil, ar2, mp, p->wtil, s);
    GemmArgs y;  // Y = W~ U^-T
    y.A = p->wtil; y.lda = mp; y.B = p->uinv; y.ldb = mp; y.C = p->kj; y.ldc = mp;  // kj is free after potrf; R~ stays in bmat
    y.M = mp; y.N = mp; y.K = mp; y.tri = TRI_KLO_BN;
    const int wks = ((int64_t)(mp / TILE) * (mp / TILE) <= 256) ? mxm_slices(mp) : 1;
    gemm_splitk(p, OP_NT, y, wks);
    GemmArgs w;  // W = U^-1 Y   (lib/fitc_gp.ml:1196-1203)
    w.A = p->uinv; w.lda = mp; w.B = p->kj; w.ldb = mp; w.C = p->wmat; w.ldc = mp;
    w.M = mp; w.N = mp; w.K = mp; w.tri = TRI_KLO_BM;
    gemm_splitk(p, OP_NN, w, wks);
    if (p->has_ms()) {
      launch_km_traces_ms(p->wmat, p->km, p->Z, p->ms, m, mp, d, p->kmpart, s);
      launch_reduce_rows(p->kmpart, nkslab, (2 * d + 2) * mp, p->kmred, 0, s);
    } else {
      launch_km_traces(p->wmat, p->km, p->Z, m, mp, d, p->kmpart, p->cp, s);
      launch_reduce_rows(p->kmpart, nkslab, (d + 2) * mp, p->kmred, 0, s);
    }
    // W_ii for the `Diag_vec / multiscale diagonal terms, into the result block
    if (wdiag) hipLaunchKernelGGL(copy_diag_kernel, dim3((mp + 255) / 256), dim3(256), 0, s, p->wmat, mp, p->wdiag);
    tstop(p);
  }
  // three transfers into pinned memory bring everything the host assembly needs: the result block (its K_m-trace and
  // diag-W parts only after a gradient evaluation), the scalar tail of the reduced exchange-1 buffer (kept in p->ar1 by
  // pass 2), and the exchange-2 buffer from its column block on (its scalar tail only after an evidence-only evaluation)
  const int64_t res_used = NSCAL + 2 + mp + (p->want_grad ? p->km_rows() * mp + (wdiag ? mp : 0) : 0);
  // -- by one launch of ship_kernel (finalize.hip): three hipMemcpyAsync calls cost 17-18 us of idle stream before them
  ShipArgs sh;
  sh.src[0] = p->res_dev; sh.dst[0] = p->res_host; sh.n[0] = res_used;
  if (!p->a1_in_scal) {
    sh.src[1] = p->ar1 + packed_upper_len(mp) + mp; sh.dst[1] = p->ex_host; sh.n[1] = A1_TAIL;
  }
  if (p->want_grad) {  // (an evidence-only evaluation reads nothing of the exchange-2 buffer)
    sh.src[2] = ar2_col; sh.dst[2] = p->ex_host + A1_TAIL; sh.n[2] = n_a2;
  }
  launch_ship(sh, s);
}

// Finish stage, second half: wait for the stream, check the factorisations, assemble l1, l2, dl/dsigma2 and the gradient
// in the reference's Hyper.get_all order on the host.  light: flags and state only (see do_finish_enqueue).
void do_finish_collect(gprhip_problem* p, gprhip_result* res, double* grad, double* coeffs, bool light = false) {
  if (p->stage != 3) {
    set_error("gprhip: finish collected before it was enqueued");
    throw HipFail{ST_STATE};
  }
  GPR_HIP(hipSetDevice(p->device));
  const int mp = p->mp, m = p->m, d = p->d;
  GPR_HIP(hipStreamSynchronize(p->stream));
  p->stage = 0;
  if (p->timer.on || p->timer.kernel) tcollect(p);
  const double* const hscal = p->res_host;
  int hinfo[2];
  std::memcpy(hinfo, p->res_host + NSCAL, sizeof hinfo);
  const double* const ht = p->res_host + NSCAL + 2;
  const double* const hkm = ht + mp;
  const double* const hwdiag = hkm + p->km_rows() * mp;
  const double* const ha1tail = p->a1_in_scal ? hscal + SC_A1TAIL : p->ex_host;
  const double* const hcol = p->ex_host + A1_TAIL;  // column block + Proj second term + scalar tail of exchange 2
  const double* const htail = hcol + p->col_rows() * mp + (int64_t)p->dbig() * d;
  if (hinfo[0] == POTRF_CHAIN_ABORT_CODE || hinfo[1] == POTRF_CHAIN_ABORT_CODE) {
    p->have_v = p->have_k = false;
    set_error("gprhip: internal error: a dependency wait of the one-launch factorisation ran into its bound "
              "(GPRHIP_POTRF_CHAIN=0 selects the step-by-step launches)");
    throw HipFail{ST_HIP_ERROR};
  }
  if (hinfo[0] != 0 || hinfo[1] != 0) {
    p->have_v = p->have_k = false;  // V came out of a failed factor
    char buf[160];
    snprintf(buf, sizeof buf,
             "Lacaml.D.potrf: leading minor of order %d of %s is not positive definite",
             hinfo[0] ? hinfo[0] : hinfo[1], hinfo[0] ? "K_m + jitter" : "B~ = I + V^T S^-1 V");
    set_error(buf);
    throw HipFail{ST_NOT_POSDEF};
  }
  p->have_model = true;
  p->have_factors = true;
  if (light) return;
  const bool mo = p->h.model_only != 0;
  const double sum_log_s = ha1tail[A1_SUMLOGS], sum_isr = ha1tail[A1_ISR], sum_isy2 = ha1tail[A1_ISY2];
  // l1: lib/fitc_gp.ml:204-208 with log|R^T R| - log|K_m| = log|B~| ; variational: :262-263
  double l1 = -0.5 * (hscal[SC_LOGDET_B] + sum_log_s + (double)p->n_total * LOG_2PI);
  if (p->h.variational) l1 += -0.5 * sum_isr;
  // l2 = -1/2 (|y~|^2 - |Q_n^T y~|^2), lib/fitc_gp.ml:290
  double l2 = mo ? 0.0 : -0.5 * (sum_isy2 - hscal[SC_BB]);
  res->l1 = l1;
  res->l2 = l2;
  res->l = l1 + l2;
  res->dl_dsigma2 = 0.0;
  res->n_hypers = 0;
  if (coeffs) std::memcpy(coeffs, ht, (size_t)m * sizeof(double));
  if (!p->want_grad) return;
  // dl/dsigma2: lib/fitc_gp.ml:1112-1119, :1187-1188
  double sumv = htail[A2_SUMV];
  res->dl_dsigma2 = -0.5 * (p->h.variational ? (sumv - htail[A2_SUMIS]) : sumv);
  // per-hyper evidence derivative, lib/fitc_gp.ml:1005-1021:
  //   dl = -1/2 (dkn_diag_term - dkm_term) - dknm_term
  double tr_wk = 0.0, tr_wkd = 0.0;
  for (int c = 0; c < m; ++c) {
    tr_wk += hkm[c];
    tr_wkd += hkm[(size_t)mp + c];
  }
  const double sumE = htail[A2_SUME], sumED = htail[A2_SUMED];
  const double scale = p->cp.inv_ell2;
  // Log_sf2: `Factor 1. on all three (lib/cov_se_iso.ml:248, :298, :302)
  const double g_sf2 = -0.5 * (p->cp.sf2 * sumv - tr_wk) - sumE;
  int64_t pos = 0;
  if (p->kind == GPRHIP_COV_SE_ISO) {
    // Log_ell: dkn_diag `Const 0.; dkm `Dense K.*D/ell^2 (diag 0); dknm `Dense (lib/cov_se_iso.ml:249-260, :303-314)
    grad[pos++] = 0.5 * scale * tr_wkd - scale * sumED;
    grad[pos++] = g_sf2;
  } else {
    grad[pos++] = g_sf2;
  }
  // Inducing_hyper {ind; dim}: dkm `Sparse_rows -> 2*scale*sum_{r!=c} W_rc K_rc (z_kr - z_kc)
  // (lib/utils.ml:196-220); dknm `Sparse_cols -> scale*sum_r (x_kr - z_kc) E_rc.  With multiscales the
  // row entries carry 1/(ms_kr + ms_kc - 1) (already inside the trace kernel) and the column 1/ms_kc
  // (lib/cov_se_fat.ml:501-513, :633-638).
  const bool msm = p->has_ms();
  const int Dp = p->has_proj() ? p->D : 0;
  for (int c = 0; c < m; ++c) {
    for (int k = 0; k < d; ++k) {
      const double zk = p->hZ[(size_t)c * d + k];
      const double dkm_half = scale * hkm[(size_t)(2 + k) * mp + c];
      double dknm = scale * (hcol[(size_t)(k + 1) * mp + c] - zk * hcol[c]);
      if (msm) dknm /= p->hMs[(size_t)c * d + k];
      grad[pos++] = dkm_half - dknm;
    }
  }
  // Proj {big_dim; small_dim} (lib/cov_se_fat.ml:429, :531, :570-596): dkm, dkn_diag `Const 0.;
  // dknm `Dense x_big,r (z_small,c - p_small,r) [/ ms_small,c] K_rc
  if (p->has_proj()) {
    const int D = p->D;
    const double* m1 = hcol + (size_t)(d + 1) * mp;       // [big][c] = sum_r x_big,r E_rc
    const double* term2 = hcol + (size_t)p->col_rows() * mp;  // [big][small]
    for (int big = 0; big < D; ++big) {
      for (int small = 0; small < d; ++small) {
        double term1 = 0.0;
        for (int c = 0; c < m; ++c) {
          double zz = p->hZ[(size_t)c * d + small];
          if (msm) zz /= p->hMs[(size_t)c * d + small];
          term1 += zz * m1[(size_t)big * mp + c];
        }
        grad[pos++] = -(term1 - term2[(size_t)big * d + small]);
      }
    }
  }
  // Log_hetero_skedasticity i (lib/cov_se_fat.ml:430-440): dkm `Diag_vec het_i e_i, the other two `Const 0.
  //   -> dl = 1/2 het_i W_ii   (lib/fitc_gp.ml:962-967)
  if (p->has_het())
    for (int i = 0; i < m; ++i) grad[pos++] = 0.5 * p->hHet[i] * hwdiag[i];
  // Log_multiscale_m05 {ind; dim} (lib/cov_se_fat.ml:441-485, :598-622), theta = log(ms - 1/2):
  //   dkm `Sparse_rows: inner_i K_i,ind for i != ind, (1/2 - ms)/(2 ms - 1) K_ind,ind on the diagonal
  //   dknm `Sparse_cols: inner_r K_r,ind with inner = (1/ms - ((p_kr - z_kc)/ms)^2) * (1/2)(1/2 - ms)
  if (msm) {
    const double* gxx = hcol + (size_t)(d + 1 + Dp) * mp;   // [k][c] = sum_r p_kr^2 E_rc
    for (int c = 0; c < m; ++c) {
      double lsum = 0.0;  // K_cc without heteroskedastic noise: exp(log_sf2 - 1/2 sum log(2 ms - 1))
      for (int k = 0; k < d; ++k) lsum += std::log(2.0 * p->hMs[(size_t)c * d + k] - 1.0);
      const double kcc = std::exp(p->cp.log_sf2 - 0.5 * lsum);
      for (int k = 0; k < d; ++k) {
        const double msk = p->hMs[(size_t)c * d + k], zk = p->hZ[(size_t)c * d + k];
        const double mh = 0.5 - msk, factor = 0.5 * mh;
        const double dkm = 2.0 * factor * hkm[(size_t)(2 + d + k) * mp + c] + hwdiag[c] * mh / (2.0 * msk - 1.0) * kcc;
        const double colE = hcol[c], gx = hcol[(size_t)(k + 1) * mp + c], g2 = gxx[(size_t)k * mp + c];
        const double dknm = factor * (colE / msk - (g2 - 2.0 * zk * gx + zk * zk * colE) / (msk * msk));
        grad[pos++] = 0.5 * dkm - dknm;
      }
    }
  }
  res->n_hypers = pos;
}

// Means.calc / Variances.calc (lib/fitc_gp.ml:418-425, :498-518) at nt test points, chunk by chunk,
// with the m x m state of the last evaluation: V_t = K_tm U^-1, Q_t = V_t R~^-1 (= K_tm R^-1).
template <typename TS>
void do_predict(gprhip_problem* p, const double* test_inputs, int64_t ld, int64_t nt, int predictive,
                double* means, double* variances) {
  if (!p->have_model || p->stage != 0) {
    set_error("gprhip_predict: no completed evaluation to predict from");
    throw HipFail{ST_STATE};
  }
  if (variances && !p->have_factors) {
    set_error("gprhip_predict: the loaded predictor has no co-variance coefficients (chol_km, r_mat)");
    throw HipFail{ST_STATE};
  }
  if (means) need_trustworthy_coeffs(p, "gprhip_predict");
  GPR_HIP(hipSetDevice(p->device));
  hipStream_t s = p->stream;
  const int mp = p->mp;
  // Test points go through in chunks of their own: the training chunk (at most the training set, padded) when that is
  // enough, else up to 131072 rows in buffers kept for later calls -- a model trained on 2000 points predicts 100 000
  // test points in one pass instead of 49 (5.4 -> 0.7 ms), each of which ends in a stream synchronisation.
  int64_t chunk = p->chunk;
  const int64_t want = std::min<int64_t>(131072, round_up(nt, TILE));
  const bool proj = p->has_proj();
  // few inducing points: a chunk is one kernel (small.hip) that keeps its n x m tiles in LDS -- it needs the row buffers
  // only, not the two chunk x mp matrices (268 MB at 131072 rows that a cached small model would pin for nothing)
  const bool small = p->small_path && !p->f32 && p->m <= 64 && mp == TILE && p->d <= 16 && !p->has_ms();
  auto release = [p](auto*& q) {  // a prediction buffer that is being regrown: freed now, not at problem destruction
    if (!q) return;
    auto it = std::find(p->allocs.begin(), p->allocs.end(), static_cast<void*>(q));
    if (it != p->allocs.end()) p->allocs.erase(it);
    (void)hipFree(q);
    q = nullptr;
  };
  if (want > chunk) {
    const int64_t rows = std::min<int64_t>(131072, std::max(want, 8 * chunk));
    if (!small && p->pred_rows < want) {
      GPR_HIP(hipStreamSynchronize(s));
      release(p->predA);
      release(p->predB);
      p->pred_rows = 0;
      p->predA = p->alloc<char>(rows * mp * p->esz);
      p->predB = p->alloc<char>(rows * mp * p->esz);
      p->pred_rows = rows;
    }
    chunk = small ? std::max(p->xt_rows >= want ? p->xt_rows : rows, p->chunk) : p->pred_rows;
  }
  if (!p->xt || p->xt_rows < chunk) {
    GPR_HIP(hipStreamSynchronize(s));
    release(p->xt);
    release(p->pt);
    release(p->prow);
    p->xt_rows = 0;
    p->xt = p->alloc<double>(chunk * p->D);
    p->pt = p->alloc<double>(chunk * p->d);
    p->prow = p->alloc<double>(3 * chunk);
    p->xt_rows = chunk;
  }
  TS* const bufA = static_cast<TS*>(chunk > p->chunk ? p->predA : p->bufA);
  TS* const bufB = static_cast<TS*>(chunk > p->chunk ? p->predB : p->bufB);
  double* rmean = p->prow;
  double* rk = p->prow + chunk;
  double* rb = p->prow + 2 * chunk;
  for (int64_t lo = 0; lo < nt; lo += chunk) {
    const int rows = (int)std::min<int64_t>(chunk, nt - lo);
    const int rows_p = (int)round_up(rows, TILE);
    GPR_HIP(hipMemcpy2DAsync(p->xt, (size_t)p->D * sizeof(double), test_inputs + lo * ld,
                             (size_t)ld * sizeof(double), (size_t)p->D * sizeof(double), (size_t)rows,
                             hipMemcpyHostToDevice, s));
    const double* pts = p->xt;
    if (proj) {
      launch_project(p->xt, rows, p->D, p->d, p->tproj, p->pt, s);
      pts = p->pt;
    }
    if (small) {  // few inducing points: the whole chunk in one kernel (small.hip)
      SmallPredictArgs a;
      a.cp = p->cp; a.pts = pts; a.Z = p->Z; a.uinv = p->uinv; a.rinv = p->rinv; a.tvec = p->tvec;
      a.rows = rows; a.m = p->m; a.mp = mp; a.d = p->d; a.add = predictive ? p->h.sigma2 : 0.0;
      a.means = means ? rmean : nullptr; a.vars = variances ? rk : nullptr;
      launch_small_predict(a, s);
      if (means) GPR_HIP(hipMemcpyAsync(means + lo, rmean, (size_t)rows * sizeof(double), hipMemcpyDeviceToHost, s));
      if (variances) GPR_HIP(hipMemcpyAsync(variances + lo, rk, (size_t)rows * sizeof(double), hipMemcpyDeviceToHost, s));
      GPR_HIP(hipStreamSynchronize(s));
      continue;
    }
    launch_cov_cross<TS>(p->cp, pts, rows, rows_p, p->Z, p->m, mp, p->d, bufA, s);
    if (means) {
      launch_row_sumsq_dot<TS>(bufA, p->tvec, rows, mp, nullptr, rmean, s);
      GPR_HIP(hipMemcpyAsync(means + lo, rmean, (size_t)rows * sizeof(double), hipMemcpyDeviceToHost, s));
    }
    if (variances) {
      GemmArgsT<TS> g;  // trsm ~side:`R chol_km
      g.A = bufA; g.lda = mp; g.B = inv_u<TS>(p); g.ldb = mp; g.C = bufB; g.ldc = mp;
      g.M = rows_p; g.N = mp; g.K = mp; g.tri = TRI_KHI_BN;
      launch_gemm(OP_NN, g, s);
      launch_row_sumsq_dot<TS>(bufB, nullptr, rows, mp, rk, nullptr, s);
      GemmArgsT<TS> q;  // trsm ~side:`R r_mat  (R = R~ U)
      q.A = bufB; q.lda = mp; q.B = inv_r<TS>(p); q.ldb = mp; q.C = bufA; q.ldc = mp;
      q.M = rows_p; q.N = mp; q.K = mp; q.tri = TRI_KHI_BN;
      launch_gemm(OP_NN, q, s);
      launch_row_sumsq_dot<TS>(bufA, nullptr, rows, mp, rb, nullptr, s);
      launch_variance_combine(rk, rb, rows, p->cp.sf2, predictive ? p->h.sigma2 : 0.0, rk, s);
      GPR_HIP(hipMemcpyAsync(variances + lo, rk, (size_t)rows * sizeof(double), hipMemcpyDeviceToHost, s));
    }
    GPR_HIP(hipStreamSynchronize(s));  // xt / row buffers are reused by the next chunk
  }
}

// Temporary device memory of one posterior call (sizes depend on the number of test points).
struct DevBuf {
  std::vector<void*> ptrs;
  template <typename T>
  T* get(int64_t count) {
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, (size_t)std::max<int64_t>(count, 1) * sizeof(T));
    if (e != hipSuccess) {
      (void)hipGetLastError();
      set_error("gprhip: device allocation failed (posterior buffers)");
      throw HipFail{ST_OOM};
    }
    ptrs.push_back(q);
    return static_cast<T*>(q);
  }
  ~DevBuf() {
    for (void* q : ptrs) (void)hipFree(q);
  }
};

// 2-norm condition estimate of A = K_m + jitter = U^T U of the current model state by power iteration on A (x <- U^T U x)
// and on A^-1 = U^-1 U^-T -- both factors are on the device, a step is two triangular matrix-vector products.  Power
// iteration approaches an extreme eigenvalue from inside the spectrum, so the estimate is a lower bound of the true
// condition number (within a small factor after 16 steps for the spectra a covariance matrix has).  Cached per evaluation.
// `enough`: a caller that only needs to know whether the estimate exceeds a threshold (the fp32-bulk guard) passes it -- the
// pivot cross-check below is itself a lower bound and comes from one small transfer, so an ill-conditioned factor is
// recognised without a single iteration step.
double condition_km(gprhip_problem* p, double enough = HUGE_VAL) {
  if (p->cond_km >= 0.0) return p->cond_km;
  if (!p->have_factors) {
    set_error("gprhip_condition: no factor of K_m on the device (evaluate, or load a predictor with co-variance coefficients)");
    throw HipFail{ST_STATE};
  }
  GPR_HIP(hipSetDevice(p->device));
  hipStream_t s = p->stream;
  const int mp = p->mp, m = p->m;
  DevBuf tmp;
  double* x0 = tmp.get<double>(3 * (int64_t)mp);
  double* const xv[2] = {x0, x0 + mp};
  double* y = x0 + 2 * (int64_t)mp;
  // Cross-check from the factor itself, first: every squared pivot U_ii^2 lies inside the spectrum of A (it is the
  // reciprocal of a diagonal entry of the inverse of a leading block, whose eigenvalues interlace A's), so
  // max U_ii^2 / min U_ii^2 is a lower bound of cond(A); the larger of the two estimates is reported.
  double pivot_ratio = 0.0;
  {
    hipLaunchKernelGGL(copy_diag_kernel, dim3((mp + 255) / 256), dim3(256), 0, s, p->umat, mp, y);
    std::vector<double> dg(mp);
    GPR_HIP(hipMemcpyAsync(dg.data(), y, (size_t)mp * sizeof(double), hipMemcpyDeviceToHost, s));
    GPR_HIP(hipStreamSynchronize(s));
    double dmax = 0.0, dmin = HUGE_VAL;
    for (int i = 0; i < m; ++i) {
      dmax = std::max(dmax, dg[i] * dg[i]);
      dmin = std::min(dmin, dg[i] * dg[i]);
    }
    if (dmin > 0.0) pivot_ratio = dmax / dmin;
    if (pivot_ratio >= enough) return pivot_ratio;  // (not cached: a later gprhip_condition wants the full estimate)
  }
  std::vector<double> h0(mp, 0.0);
  for (int i = 0; i < m; ++i) h0[i] = 1.0 + 0.5 * std::sin(1.0 + 0.37 * i);  // the padding (a decoupled identity block) stays 0
  // Rounds of 16 steps until both Rayleigh quotients have settled (relative change < 1e-2 over a round; at most 16 rounds:
  // lambda_max(A^-1) converges slowly when K_m + jitter has a cluster of small eigenvalues, and a fixed 16 steps could pass
  // exactly the coefficients the fp32 guard exists to refuse).  The two iterations advance side by side: one transfer and
  // one host synchronisation per round for both.
  double lam[2] = {0.0, 0.0}, prev[2] = {0.0, 0.0};
  bool settled[2] = {false, false};
  for (int which = 0; which < 2; ++which) {
    GPR_HIP(hipMemcpyAsync(xv[which], h0.data(), (size_t)mp * sizeof(double), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(normalize_kernel, dim3(1), dim3(256), 0, s, xv[which], mp, p->scal + SC_LMAX + which);
  }
  for (int round = 0; round < 16 && !(settled[0] && settled[1]); ++round) {
    for (int which = 0; which < 2; ++which) {
      if (settled[which]) continue;
      const double* F = which == 0 ? p->umat : p->uinv;
      double* x = xv[which];
      for (int it = 0; it < 16; ++it) {
        if (which == 0) {  // x <- U^T (U x)
          launch_triu_matvec(F, mp, x, y, 0, s);
          launch_triu_matvec(F, mp, y, x, 1, s);
        } else {           // x <- U^-1 (U^-T x)
          launch_triu_matvec(F, mp, x, y, 1, s);
          launch_triu_matvec(F, mp, y, x, 0, s);
        }
        hipLaunchKernelGGL(normalize_kernel, dim3(1), dim3(256), 0, s, x, mp, p->scal + SC_LMAX + which);
      }
    }
    GPR_HIP(hipMemcpyAsync(lam, p->scal + SC_LMAX, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
    GPR_HIP(hipStreamSynchronize(s));
    for (int which = 0; which < 2; ++which) {
      if (settled[which]) continue;
      if (round > 0 && std::fabs(lam[which] - prev[which]) <= 1e-2 * std::fabs(lam[which])) settled[which] = true;
      prev[which] = lam[which];
    }
  }
  p->cond_km = std::max(pivot_ratio, lam[0] * lam[1]);
  return p->cond_km;
}

// fp32-bulk problems: the mean coefficients t = U^-1 t~ inherit cond(K_m + jitter) times the fp32 unit roundoff of the
// n x m operands (measured: profiles/r03_f32_sweep.txt -- useless from cond ~ 1e6 on); their consumers refuse them then.
void need_trustworthy_coeffs(gprhip_problem* p, const char* who) {
  if (!p->f32 || p->f32_coeff_tol <= 0.0 || !p->have_factors) return;
  // (an estimate beyond twice the threshold settles the question: the pivot ratio alone may then answer, without iterating)
  const double cond = condition_km(p, 2.0 * p->f32_coeff_tol / 5.9604644775390625e-8);
  const double bound = cond * 5.9604644775390625e-8;
  if (bound > p->f32_coeff_tol) {
    char buf[320];
    snprintf(buf, sizeof buf,
             "%s: the mean coefficients of this fp32-bulk evaluation are not trustworthy: cond(K_m + jitter) ~ %.3g, "
             "error bound %.3g > %.3g (GPRHIP_F32_COEFF_TOL); evaluate with GPRHIP_F64 for coefficients / predictions "
             "(log evidence and gradient are unaffected)", who, cond, bound, p->f32_coeff_tol);
    set_error(buf);
    throw HipFail{GPRHIP_EPRECISION};
  }
}

void need_factors(gprhip_problem* p, const char* who) {
  if (!p->have_factors) {
    set_error(std::string(who) + ": the loaded predictor has no co-variance coefficients (chol_km, r_mat)");
    throw HipFail{ST_STATE};
  }
}

void need_model(gprhip_problem* p, const char* who) {
  if (!p->have_model || p->stage != 0) {
    set_error(std::string(who) + ": no completed evaluation to work from");
    throw HipFail{ST_STATE};
  }
}

// Trained.calc_means (lib/fitc_gp.ml:296-297) over the resident training inputs, and the residual sums
// Stats.calc needs (lib/fitc_gp.ml:353-373): sums = { sse, sum |y-mean|, max |y-mean|, sum y^2 }.
template <typename TS>
void do_train_stats(gprhip_problem* p, double* means, double* sums) {
  need_model(p, "gprhip_train_stats");
  if (!p->have_targets || p->h.model_only) {
    set_error("gprhip_train_stats: the last evaluation had no targets");
    throw HipFail{ST_STATE};
  }
  need_trustworthy_coeffs(p, "gprhip_train_stats");
  GPR_HIP(hipSetDevice(p->device));
  hipStream_t s = p->stream;
  DevBuf tmp;
  double* rmean = tmp.get<double>(p->chunk);
  int64_t nblocks = 0;
  for (int c = 0; c < p->nchunks; ++c) nblocks += residual_stat_blocks((int)p->rows_of(c));
  double* part = tmp.get<double>(nblocks * 4);
  TS* const K = static_cast<TS*>(p->bufA);
  int64_t b0 = 0;
  for (int c = 0; c < p->nchunks; ++c) {
    const int rows = (int)p->rows_of(c);
    const int64_t lo = (int64_t)c * p->chunk;
    cov_chunk<TS>(p, c, K);
    launch_row_sumsq_dot<TS>(K, p->tvec, rows, p->mp, nullptr, rmean, s);
    launch_residual_stats(p->y + lo, rmean, rows, part + b0 * 4, s);
    b0 += residual_stat_blocks(rows);
    if (means)
      GPR_HIP(hipMemcpyAsync(means + lo, rmean, (size_t)rows * sizeof(double), hipMemcpyDeviceToHost, s));
  }
  std::vector<double> hp((size_t)nblocks * 4);
  GPR_HIP(hipMemcpyAsync(hp.data(), part, hp.size() * sizeof(double), hipMemcpyDeviceToHost, s));
  GPR_HIP(hipStreamSynchronize(s));
  double sse = 0.0, sad = 0.0, mad = 0.0, sy2 = 0.0;
  for (int64_t b = 0; b < nblocks; ++b) {
    sse += hp[b * 4 + 0];
    sad += hp[b * 4 + 1];
    mad = std::max(mad, hp[b * 4 + 2]);
    sy2 += hp[b * 4 + 3];
  }
  sums[0] = sse; sums[1] = sad; sums[2] = mad; sums[3] = sy2;
}

// FITC_covariances.calc / FIC_covariances.calc (lib/fitc_gp.ml:585-599, :617-627) at nt test points, fp64:
//   V_t = K_tm U^-1, Q_t = K_tm R^-1 = V_t R~^-1
//   FITC: K_tt - V_t V_t^T + Q_t Q_t^T          FIC: Q_t Q_t^T + diag(k_tt - rowsum(K_tm.^2))
// (the FIC diagonal is what the reference computes at :620 -- from K_tm itself, not from V_t).
void do_covariances(gprhip_problem* p, const double* test_inputs, int64_t ld, int64_t nt, int kind,
                    int predictive, double* cov) {
  need_model(p, "gprhip_covariances");
  need_factors(p, "gprhip_covariances");
  GPR_HIP(hipSetDevice(p->device));
  hipStream_t s = p->stream;
  const int mp = p->mp;
  const int np = (int)round_up(nt, TILE);
  DevBuf tmp;
  double* xt = tmp.get<double>(nt * p->D);
  double* Kt = tmp.get<double>((int64_t)np * mp);
  double* Vt = tmp.get<double>((int64_t)np * mp);
  double* C = tmp.get<double>((int64_t)np * np);
  double* rv = tmp.get<double>(2 * (int64_t)np);
  GPR_HIP(hipMemcpy2DAsync(xt, (size_t)p->D * sizeof(double), test_inputs, (size_t)ld * sizeof(double),
                           (size_t)p->D * sizeof(double), (size_t)nt, hipMemcpyHostToDevice, s));
  const double* pts = xt;
  if (p->has_proj()) {
    double* pt = tmp.get<double>(nt * p->d);
    launch_project(xt, nt, p->D, p->d, p->tproj, pt, s);
    pts = pt;
  }
  launch_cov_cross<double>(p->cp, pts, (int)nt, np, p->Z, p->m, mp, p->d, Kt, s);
  GemmArgs g;  // trsm ~side:`R chol_km
  g.A = Kt; g.lda = mp; g.B = p->uinv; g.ldb = mp; g.C = Vt; g.ldc = mp;
  g.M = np; g.N = mp; g.K = mp; g.tri = TRI_KHI_BN;
  launch_gemm(OP_NN, g, s);
  if (kind == 0) {
    // Inputs.calc_upper: the plain kernel between the (projected) test points, no multiscales
    // (lib/cov_se_iso.ml:124, lib/cov_se_fat.ml:221 -> calc_upper_vanilla :85-100)
    CovParams plain = p->cp;
    plain.ms = nullptr;
    launch_cov_cross<double>(plain, pts, (int)nt, np, pts, (int)nt, np, p->d, C, s);
    GemmArgs a;  // syrk ~alpha:-1 tmp ~c:covariances
    a.A = Vt; a.lda = mp; a.B = Vt; a.ldb = mp; a.C = C; a.ldc = np;
    a.M = np; a.N = np; a.K = mp; a.alpha = -1.0; a.beta = 1.0;
    launch_gemm(OP_NT, a, s);
  } else {
    launch_row_sumsq_dot<double>(Kt, nullptr, (int)nt, mp, rv, nullptr, s);
    launch_const_minus(rv, (int)nt, p->cp.sf2, rv + np, s);
  }
  GemmArgs q;  // trsm ~side:`R r_mat  (R = R~ U)
  q.A = Vt; q.lda = mp; q.B = p->rinv; q.ldb = mp; q.C = Kt; q.ldc = mp;
  q.M = np; q.N = mp; q.K = mp; q.tri = TRI_KHI_BN;
  launch_gemm(OP_NN, q, s);
  GemmArgs b;  // syrk tmp ~c:covariances
  b.A = Kt; b.lda = mp; b.B = Kt; b.ldb = mp; b.C = C; b.ldc = np;
  b.M = np; b.N = np; b.K = mp; b.alpha = 1.0; b.beta = kind == 0 ? 1.0 : 0.0;
  launch_gemm(OP_NT, b, s);
  const double add = predictive ? p->h.sigma2 : 0.0;  // get ?predictive, :549-559
  if (kind != 0 || add != 0.0) launch_add_diag(C, np, (int)nt, kind != 0 ? rv + np : nullptr, add, s);
  // symmetric: row-major == column-major
  GPR_HIP(hipMemcpy2DAsync(cov, (size_t)nt * sizeof(double), C, (size_t)np * sizeof(double),
                           (size_t)nt * sizeof(double), (size_t)nt, hipMemcpyDeviceToHost, s));
  GPR_HIP(hipStreamSynchronize(s));
}

// Common_cov_sampler.calc + samples (lib/fitc_gp.ml:656-697): cov_chol = chol(cov + add_diag I + jitter I)
// (upper), samples[:, j] = means + cov_chol^T z[:, j].  z is supplied by the caller (the reference draws
// it from GSL's ziggurat generator).
void do_cov_samples(gprhip_problem* p, const double* cov, int64_t ld, int64_t nt, double add_diag,
                    double jitter, const double* means, const double* z, int64_t ns, double* samples) {
  GPR_HIP(hipSetDevice(p->device));
  hipStream_t s = p->stream;
  const int np = (int)round_up(nt, TILE);
  const int nsp = (int)round_up(ns, TILE);
  DevBuf tmp;
  double* raw = tmp.get<double>(nt * nt);
  double* A = tmp.get<double>((int64_t)np * np);
  double* dinv = tmp.get<double>((int64_t)np * TILE);
  double* Zd = tmp.get<double>((int64_t)nsp * np);
  double* S = tmp.get<double>((int64_t)nsp * np);
  double* mu = tmp.get<double>(nt);
  int* info = tmp.get<int>(1);
  GPR_HIP(hipMemcpy2DAsync(raw, (size_t)nt * sizeof(double), cov, (size_t)ld * sizeof(double),
                           (size_t)nt * sizeof(double), (size_t)nt, hipMemcpyHostToDevice, s));
  GPR_HIP(hipMemsetAsync(info, 0, sizeof(int), s));
  launch_sym_from_upper(raw, nt, (int)nt, A, np, add_diag + jitter, s);
  potrf_upper_n(s, A, np, dinv, info, p->engine_steps, (int)nt);
  int hinfo = 0;
  GPR_HIP(hipMemcpyAsync(&hinfo, info, sizeof(int), hipMemcpyDeviceToHost, s));
  // z: Fortran nt x ns == row-major [ns][nt]
  GPR_HIP(hipMemsetAsync(Zd, 0, (size_t)nsp * np * sizeof(double), s));
  GPR_HIP(hipMemcpy2DAsync(Zd, (size_t)np * sizeof(double), z, (size_t)nt * sizeof(double),
                           (size_t)nt * sizeof(double), (size_t)ns, hipMemcpyHostToDevice, s));
  GPR_HIP(hipMemcpyAsync(mu, means, (size_t)nt * sizeof(double), hipMemcpyHostToDevice, s));
  GemmArgs g;  // trmm ~transa:`T cov_chol samples  ==  (z^T U)^T
  g.A = Zd; g.lda = np; g.B = A; g.ldb = np; g.C = S; g.ldc = np;
  g.M = nsp; g.N = np; g.K = np; g.tri = TRI_KHI_BN;
  launch_gemm(OP_NN, g, s);
  launch_add_row_vector(S, np, (int)ns, (int)nt, mu, s);
  GPR_HIP(hipMemcpy2DAsync(samples, (size_t)nt * sizeof(double), S, (size_t)np * sizeof(double),
                           (size_t)nt * sizeof(double), (size_t)ns, hipMemcpyDeviceToHost, s));
  GPR_HIP(hipStreamSynchronize(s));
  if (hinfo != 0) {
    set_error("Cov_sampler.calc: potrf: leading minor of order " + std::to_string(hinfo) +
              " is not positive definite");
    throw HipFail{ST_NOT_POSDEF};
  }
}

// Row-major padded upper factor -> Fortran m x m (upper triangle, zeros below) on the host.
void fetch_upper_fortran(gprhip_problem* p, const double* dev, double* out) {
  const int mp = p->mp, m = p->m;
  std::vector<double> h((size_t)mp * mp);
  GPR_HIP(hipMemcpyAsync(h.data(), dev, h.size() * sizeof(double), hipMemcpyDeviceToHost, p->stream));
  GPR_HIP(hipStreamSynchronize(p->stream));
  for (int c = 0; c < m; ++c)
    for (int r = 0; r < m; ++r) out[(size_t)c * m + r] = (r <= c) ? h[(size_t)r * mp + c] : 0.0;
}

// Fortran m x m upper factor on the host -> row-major padded mp x mp on the device (zeros below, identity padding).
void upload_upper_fortran(gprhip_problem* p, const double* in, double* dev) {
  const int mp = p->mp, m = p->m;
  std::vector<double> h((size_t)mp * mp, 0.0);
  for (int r = 0; r < mp; ++r)
    for (int c = r; c < mp; ++c)
      h[(size_t)r * mp + c] = (c < m) ? in[(size_t)c * m + r] : (r == c ? 1.0 : 0.0);
  GPR_HIP(hipMemcpyAsync(dev, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice, p->stream));
  GPR_HIP(hipStreamSynchronize(p->stream));  // h goes out of scope
}

// Model.calc_co_variance_coeffs (lib/fitc_gp.ml:240): (chol_km, r_mat) = (U, R~ U) of the last evaluation.
void do_co_variance_coeffs(gprhip_problem* p, double* chol_km, double* r_mat) {
  need_model(p, "gprhip_co_variance_coeffs");
  need_factors(p, "gprhip_co_variance_coeffs");
  GPR_HIP(hipSetDevice(p->device));
  const int mp = p->mp;
  if (chol_km) fetch_upper_fortran(p, p->umat, chol_km);
  if (r_mat) {
    GemmArgs g;  // R = R~ U, both upper triangular
    g.A = p->bmat; g.lda = mp; g.B = p->umat; g.ldb = mp; g.C = p->wmat; g.ldc = mp;
    g.M = mp; g.N = mp; g.K = mp; g.tri = TRI_BAND; g.upper_only = 1;
    launch_gemm(OP_NN, g, p->stream);
    fetch_upper_fortran(p, p->wmat, r_mat);
  }
}

// inv of an upper factor that is already on the device: per-block inverses, then the recursive-doubling joins
void trtri_of_factor(gprhip_problem* p, double* U, double* X) {
  for (int j = 0; j < p->mp / TILE; ++j)
    launch_potrf_diag_flags(U, p->mp, j, p->dinv + (int64_t)j * TILE * TILE, p->info, 1, p->stream, p->m);
  trtri_upper(p, U, X, p->wmat);
}

// Mean_predictor.calc + Co_variance_predictor.calc (lib/fitc_gp.ml:386-391, :446-447): install the predictor
// state of a saved model -- kernel, inducing points, mean coefficients, (chol_km, r_mat) -- without an evaluation.
void do_load_predictor(gprhip_problem* p, const gprhip_hypers* h, const double* coeffs, const double* chol_km,
                       const double* r_mat) {
  GPR_HIP(hipSetDevice(p->device));
  hipStream_t s = p->stream;
  const int mp = p->mp, m = p->m;
  const int64_t mm = (int64_t)mp * mp;
  p->have_model = p->have_factors = p->have_v = p->have_k = false;
  p->cond_km = -1.0;
  upload_hypers(p, h);
  std::vector<double> t(mp, 0.0);
  if (coeffs) std::memcpy(t.data(), coeffs, (size_t)m * sizeof(double));
  GPR_HIP(hipMemcpyAsync(p->tvec, t.data(), (size_t)mp * sizeof(double), hipMemcpyHostToDevice, s));
  GPR_HIP(hipMemsetAsync(p->info, 0, 2 * sizeof(int), s));
  p->have_factors = chol_km != nullptr;
  if (chol_km) {
    upload_upper_fortran(p, chol_km, p->umat);
    trtri_of_factor(p, p->umat, p->uinv);
    upload_upper_fortran(p, r_mat, p->binv);       // R, scratch
    trtri_of_factor(p, p->binv, p->wtil);          // R^-1
    GemmArgs g;  // R~^-1 = U R^-1  (R = R~ U)
    g.A = p->umat; g.lda = mp; g.B = p->wtil; g.ldb = mp; g.C = p->rinv; g.ldc = mp;
    g.M = mp; g.N = mp; g.K = mp; g.tri = TRI_BAND; g.upper_only = 1;
    GPR_HIP(hipMemsetAsync(p->rinv, 0, (size_t)mm * sizeof(double), s));
    launch_gemm(OP_NN, g, s);
    GemmArgs b;  // R~ = R U^-1, kept for a later export
    b.A = p->binv; b.lda = mp; b.B = p->uinv; b.ldb = mp; b.C = p->bmat; b.ldc = mp;
    b.M = mp; b.N = mp; b.K = mp; b.tri = TRI_BAND; b.upper_only = 1;
    GPR_HIP(hipMemsetAsync(p->bmat, 0, (size_t)mm * sizeof(double), s));
    launch_gemm(OP_NN, b, s);
    if (p->f32) {
      launch_to_float(p->uinv, p->uinv_f, mm, s);
      launch_to_float(p->rinv, p->rinv_f, mm, s);
    }
  }
  GPR_HIP(hipStreamSynchronize(s));
  p->stage = 0;
  p->have_model = true;
  p->h.model_only = coeffs ? 0 : 1;
}

void tdrop(gprhip_problem* p) {  // a failed evaluation leaves its stage timers behind
  if (!p) return;
  for (auto& e : p->timer.ev) {
    hipEventDestroy(e.second.first);
    hipEventDestroy(e.second.second);
  }
  p->timer.ev.clear();
}

template <typename F>
int guarded(F&& f, gprhip_problem* p = nullptr) {
  try {
    f();
    return GPRHIP_OK;
  } catch (const HipFail& e) {
    tdrop(p);
    return e.status;
  } catch (const std::bad_alloc&) {
    set_error("gprhip: host allocation failed");
    return GPRHIP_EOOM;
  } catch (...) {
    set_error("gprhip: unexpected C++ exception");
    return GPRHIP_EHIP;
  }
}

}  // namespace

namespace gprhip {
double* problem_ar1(gprhip_problem* p) { return p->ar1; }
double* problem_ar2(gprhip_problem* p) { return p->ar2; }
hipStream_t problem_hip_stream(gprhip_problem* p) { return p->stream; }
int problem_device(const gprhip_problem* p) { return p->device; }
int64_t problem_rows(const gprhip_problem* p) { return p->n; }
int problem_finish_enqueue(gprhip_problem* p, const double* d_ar2, int light) {
  return guarded([&] { do_finish_enqueue(p, d_ar2, light != 0); }, p);
}
int problem_finish_collect(gprhip_problem* p, gprhip_result* res, double* grad, double* coeffs, int light) {
  return guarded([&] { do_finish_collect(p, res, grad, coeffs, light != 0); }, p);
}
}  // namespace gprhip

extern "C" {

const char* gprhip_last_error(void) { return last_error().c_str(); }
const char* gprhip_version(void) { return "gprhip 0.4 (gfx950)"; }

int gprhip_device_count(int* count) {
  return guarded([&] {
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) c = 0;
    *count = c;
  });
}

int gprhip_problem_create(int device, int cov_kind, int64_t n, int D, int d, int m, int64_t chunk_rows,
                          gprhip_problem** out) {
  return gprhip_problem_create_ex(device, cov_kind, GPRHIP_F64, n, D, d, m, chunk_rows, out);
}

int gprhip_problem_create_ex(int device, int cov_kind, int precision, int64_t n, int D, int d, int m,
                             int64_t chunk_rows, gprhip_problem** out) {
  return guarded([&] {
    if (precision != GPRHIP_F64 && precision != GPRHIP_F32_BULK) {
      set_error("gprhip_problem_create_ex: unknown precision");
      throw HipFail{ST_BAD_ARG};
    }
    if (!out || n < 1 || D < 1 || d < 1 || m < 1 ||
        (cov_kind != GPRHIP_COV_SE_ISO && cov_kind != GPRHIP_COV_SE_FAT) ||
        (cov_kind == GPRHIP_COV_SE_ISO && d != D)) {
      set_error("gprhip_problem_create: invalid arguments (need n,D,d,m >= 1, d == D for Cov_se_iso)");
      throw HipFail{ST_BAD_ARG};
    }
    check_single_hip_runtime("gprhip_problem_create");
    GPR_HIP(hipSetDevice(device));
    gemm_init();
    auto* p = new gprhip_problem();
    *out = p;
    p->device = device; p->kind = cov_kind; p->n = n; p->D = D; p->d = d; p->m = m;
    p->f32 = (precision == GPRHIP_F32_BULK);
    p->esz = p->f32 ? 4 : 8;
    p->mp = (int)round_up(m, TILE);
    const Sizing z = problem_sizing(n, p->mp, chunk_rows, !p->f32);
    const int64_t chunk = z.chunk;
    p->chunk = chunk;
    p->nchunks = z.nchunks;
    p->slice_rows = z.slice_rows;
    p->kslices = z.kslices;
    {  // the whole resident set of the problem against what the device has free, BEFORE anything is allocated: a shard that
       // cannot hold its V store fails here, by name and with the figures, not in the first evaluation's hipMalloc
      gprhip_memory_plan_t plan;
      memory_plan(cov_kind, precision, n, D, d, m, chunk_rows, &plan);
      size_t free_b = 0, total_b = 0;
      GPR_HIP(hipMemGetInfo(&free_b, &total_b));
      const bool verbose = getenv("GPRHIP_VERBOSE") && atoi(getenv("GPRHIP_VERBOSE")) != 0;
      if (verbose)
        fprintf(stderr, "gprhip: device %d: problem n=%lld m=%d d=%d (%s): resident %.3f GB = V %.3f + chunk buffers %.3f + "
                        "split-K slices %.3f + inputs %.3f + rows %.3f + m x m %.3f + rest %.3f  (optional K_nm copy %.3f); "
                        "free %.3f of %.3f GB\n", device, (long long)n, m, d, p->f32 ? "fp32-bulk" : "fp64",
                plan.total / 1e9, plan.v_store / 1e9, plan.chunk_buffers / 1e9, plan.slices / 1e9, plan.inputs / 1e9,
                plan.row_vectors / 1e9, plan.mxm / 1e9, plan.rest / 1e9, plan.k_store_optional / 1e9, free_b / 1e9, total_b / 1e9);
      if ((uint64_t)plan.total > (uint64_t)free_b) {
        char buf[320];
        snprintf(buf, sizeof buf, "gprhip_problem_create: the problem needs %.1f GB on device %d (V = K_nm U^-1 of %lld rows x %d: "
                 "%.1f GB, split-K slices %.1f GB, chunk buffers %.1f GB) and %.1f GB are free of %.1f GB: shard the rows over more "
                 "devices (gprhip_ctx_create) or use the fp32-bulk mode", plan.total / 1e9, device, (long long)n, p->mp,
                 plan.v_store / 1e9, plan.slices / 1e9, plan.chunk_buffers / 1e9, free_b / 1e9, total_b / 1e9);
        set_error(buf);
        delete p;
        *out = nullptr;
        throw HipFail{ST_OOM};
      }
    }
    if (const char* e = getenv("GPRHIP_TIMING")) {
      p->timer.on = atoi(e) >= 2;
      p->timer.kernel = atoi(e) >= 1;
    }
    if (const char* e = getenv("GPRHIP_TILE_ORDER")) p->tile_order = atoi(e);
    if (const char* e = getenv("GPRHIP_GRAD_SCALAR")) p->grad_scalar = atoi(e);
    if (const char* e = getenv("GPRHIP_K_RESIDENT")) p->k_resident = atoi(e);
    if (const char* e = getenv("GPRHIP_W_AS_WS")) p->w_as_ws = atoi(e);
    if (const char* e = getenv("GPRHIP_SMALL_PATH")) p->small_path = atoi(e);
    if (const char* e = getenv("GPRHIP_MID_PATH")) p->mid_path = atoi(e);
    if (const char* e = getenv("GPRHIP_MID_GRAM")) p->mid_gram = atoi(e);
    if (const char* e = getenv("GPRHIP_F32_COEFF_TOL")) p->f32_coeff_tol = atof(e);
    if (const char* e = getenv("GPRHIP_MERGED_X")) p->merged_x_mode = atoi(e);
    if (const char* e = getenv("GPRHIP_POTRF_ENGINE")) p->engine_steps = atoi(e) != 0;
#ifdef GPRHIP_LAB  // the measured-slower variants (DESIGN sections 4, 14): switches of the lab build only
    if (const char* e = getenv("GPRHIP_POTRF_CHAIN")) p->potrf_chain_mode = atoi(e);
    if (const char* e = getenv("GPRHIP_COV_OVERLAP")) p->cov_overlap = atoi(e);
#endif
    GPR_HIP(hipStreamCreate(&p->stream));
    GPR_HIP(hipStreamCreate(&p->stream2));
    for (int k = 0; k < 2; ++k) {
      GPR_HIP(hipEventCreateWithFlags(&p->ev_cov[k], hipEventDisableTiming));
      GPR_HIP(hipEventCreateWithFlags(&p->ev_vdone[k], hipEventDisableTiming));
    }
#ifdef GPRHIP_LAB
    if (p->mp >= 3 * TILE && getenv("GPRHIP_POTRF_LOOKAHEAD") && atoi(getenv("GPRHIP_POTRF_LOOKAHEAD")) > 0) {  // (A/B runs only)
      p->potrf_aux.min_rest = atoi(getenv("GPRHIP_POTRF_LOOKAHEAD"));
      GPR_HIP(hipStreamCreate(&p->potrf_aux.side));
      for (int k = 0; k < 2; ++k) {
        GPR_HIP(hipEventCreateWithFlags(&p->potrf_aux.ev_panel[k], hipEventDisableTiming));
        GPR_HIP(hipEventCreateWithFlags(&p->potrf_aux.ev_rest[k], hipEventDisableTiming));
      }
    }
#endif
    GPR_HIP(hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming));
    GPR_HIP(hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming));
    GPR_HIP(hipEventCreateWithFlags(&p->ev_rf, hipEventDisableTiming));
    GPR_HIP(hipEventCreateWithFlags(&p->ev_binv, hipEventDisableTiming));
    const int mp = p->mp;
    const int64_t mm = (int64_t)mp * mp;
    const int64_t npad = (int64_t)p->nchunks * chunk;
    p->X = p->alloc<double>(n * D);
    p->y = p->alloc<double>(npad);
    if (cov_kind == GPRHIP_COV_SE_FAT) p->P = p->alloc<double>(n * d);
    {  // the per-evaluation parameter block and its pinned host mirror (upload_hypers)
      const bool fat = cov_kind == GPRHIP_COV_SE_FAT;
      const int64_t o_shift = (int64_t)mp * d, o_tproj = o_shift + 64, o_het = o_tproj + (fat ? round_up((int64_t)D * d, 2) : 0),
                    o_ms = o_het + (fat ? mp : 0);
      p->hy_len = o_ms + (fat ? (int64_t)mp * d : 0);
      p->hy_dev = p->alloc<double>(p->hy_len);
      GPR_HIP(hipHostMalloc(reinterpret_cast<void**>(&p->hy_host), (size_t)p->hy_len * sizeof(double), hipHostMallocDefault));
      GPR_HIP(hipEventCreateWithFlags(&p->ev_hy, hipEventDisableTiming));
      p->Z = p->hy_dev; p->hZ = p->hy_host;
      p->zshift = p->hy_dev + o_shift; p->hShift = p->hy_host + o_shift;
      if (fat) {
        p->tproj = p->hy_dev + o_tproj; p->hTproj = p->hy_host + o_tproj;
        p->het = p->hy_dev + o_het; p->hHet = p->hy_host + o_het;
        p->ms = p->hy_dev + o_ms; p->hMs = p->hy_host + o_ms;
      }
    }
    p->km = p->alloc<double>(mm); p->kj = p->alloc<double>(mm); p->umat = p->alloc<double>(mm);
    p->uinv = p->alloc<double>(mm); p->bmat = p->alloc<double>(mm);
    p->rinv = p->alloc<double>(mm); p->binv = p->alloc<double>(mm); p->wtil = p->alloc<double>(mm);
    p->rfinv = p->alloc<double>(mm);
    p->wmat = p->alloc<double>(mm);
    p->tmp = p->alloc<double>((int64_t)mp * TILE);
    p->dinv = p->alloc<double>((int64_t)(mp / TILE) * TILE * TILE);
    p->bvec = p->alloc<double>(mp); p->ttil = p->alloc<double>(mp);
    {  // the result block and its pinned mirror; the tails of the exchange buffers land in ex_host (do_finish_enqueue)
      p->res_len = NSCAL + 2 + mp + p->km_rows() * mp + mp;
      p->ex_len = A1_TAIL + p->col_rows() * mp + (int64_t)p->dbig() * d + A2_TAIL;
      p->res_dev = p->alloc<double>(p->res_len + p->ex_len);
      p->ex_dev = p->res_dev + p->res_len;
      p->scal = p->res_dev;
      p->info = reinterpret_cast<int*>(p->res_dev + NSCAL);
      p->tvec = p->res_dev + NSCAL + 2;
      p->kmred = p->tvec + mp;
      p->wdiag = p->kmred + p->km_rows() * mp;
      GPR_HIP(hipHostMalloc(reinterpret_cast<void**>(&p->res_host), (size_t)(p->res_len + p->ex_len) * sizeof(double),
                            hipHostMallocPortable | hipHostMallocMapped));  // (kernels of this device write into it)
      p->ex_host = p->res_host + p->res_len;
    }
    p->r = p->alloc<double>(npad); p->is = p->alloc<double>(npad); p->yis = p->alloc<double>(npad);
    p->w = p->alloc<double>(npad); p->v = p->alloc<double>(npad);
    if (cov_kind == GPRHIP_COV_SE_FAT) {
      p->es = p->alloc<double>(npad);
      p->projpart = p->alloc<double>(((chunk + 255) / 256) * (int64_t)D * d);
    }
    p->rp1 = p->alloc<double>(chunk * 4 * (mp / TILE)); p->rp2 = p->alloc<double>(chunk * 4 * (mp / TILE));
    p->bufA = p->alloc<char>(chunk * mp * p->esz); p->bufB = p->alloc<char>(chunk * mp * p->esz);
    if (p->f32) {
      p->uinv_f = p->alloc<float>(mm);
      p->rinv_f = p->alloc<float>(mm);
      p->rfinv_f = p->alloc<float>(mm);
      p->is_f = p->alloc<float>(npad); p->yis_f = p->alloc<float>(npad); p->v_f = p->alloc<float>(npad);
    }
    p->slices_bytes = (int64_t)p->kslices * mm * p->esz;
    p->slices = p->alloc<char>(p->slices_bytes);
    p->rowpart = p->alloc<double>((int64_t)pass1_row_blocks((int)chunk) * 4);
    p->gemvpart = p->alloc<double>((int64_t)p->kslices * mp);  // c~ partials, one row per k-slice
    p->grad_slab = pick_grad_slab((int)p->col_rows(), n, chunk, mp);
    const int64_t gslab = p->grad_slab;
    const int64_t nslab = (chunk + gslab - 1) / gslab;
    p->colpart = p->alloc<double>(nslab * p->col_rows() * mp);
    p->scalpart = p->alloc<double>(nslab * (mp / TILE) * 2);
    p->kmpart = p->alloc<double>((int64_t)((m + km_slab_rows(m) - 1) / km_slab_rows(m)) * p->km_rows() * mp);
    p->ar1 = p->alloc<double>(gprhip_ar1_len(p));
    p->ar2 = p->alloc<double>(gprhip_ar2_len(p));
    {
      gprhip_memory_plan_t plan;
      memory_plan(cov_kind, precision, n, D, d, m, chunk_rows, &plan);
      p->planned_bytes = plan.total;
      if (getenv("GPRHIP_VERBOSE") && atoi(getenv("GPRHIP_VERBOSE")) != 0)
        fprintf(stderr, "gprhip: device %d: allocated %.3f GB at creation, %.3f GB planned without the V store\n", device,
                p->alloc_bytes / 1e9, (plan.total - plan.v_store) / 1e9);
    }
    GPR_HIP(hipMemsetAsync(p->y, 0, (size_t)npad * sizeof(double), p->stream));
    // R^-1 is written on its upper tiles only; the fp32 conversion reads the whole square
    GPR_HIP(hipMemsetAsync(p->rfinv, 0, (size_t)mm * sizeof(double), p->stream));
    GPR_HIP(hipStreamSynchronize(p->stream));
  });
}

void gprhip_problem_destroy(gprhip_problem* p) {
  if (!p) return;
  hipSetDevice(p->device);
  if (p->stream) {
    hipStreamSynchronize(p->stream);
    hipStreamDestroy(p->stream);
  }
  if (p->stream2) {
    hipStreamSynchronize(p->stream2);
    hipStreamDestroy(p->stream2);
  }
  if (p->ev_hy) hipEventDestroy(p->ev_hy);
  if (p->hy_host) hipHostFree(p->hy_host);
  if (p->res_host) hipHostFree(p->res_host);
  if (p->ev_fork) hipEventDestroy(p->ev_fork);
  if (p->ev_join) hipEventDestroy(p->ev_join);
  if (p->ev_rf) hipEventDestroy(p->ev_rf);
  if (p->ev_binv) hipEventDestroy(p->ev_binv);
  for (int k = 0; k < 2; ++k) {
    if (p->ev_cov[k]) hipEventDestroy(p->ev_cov[k]);
    if (p->ev_vdone[k]) hipEventDestroy(p->ev_vdone[k]);
  }
  if (p->potrf_aux.side) {
    hipStreamSynchronize(p->potrf_aux.side);
    hipStreamDestroy(p->potrf_aux.side);
    for (int k = 0; k < 2; ++k) {
      hipEventDestroy(p->potrf_aux.ev_panel[k]);
      hipEventDestroy(p->potrf_aux.ev_rest[k]);
    }
  }
  if (p->timer.k0) {
    hipEventDestroy(p->timer.k0);
    hipEventDestroy(p->timer.k1);
  }
  for (void* a : p->allocs) hipFree(a);
#ifdef GPRHIP_LAB
  potrf_chain_destroy(p->chain);
#endif
  delete p;
}

int gprhip_set_inputs(gprhip_problem* p, const double* inputs, int64_t ld) {
  return guarded([&] {
    if (!p || !inputs || ld < p->D) {
      set_error("gprhip_set_inputs: invalid arguments");
      throw HipFail{ST_BAD_ARG};
    }
    GPR_HIP(hipSetDevice(p->device));
    GPR_HIP(hipMemcpy2DAsync(p->X, (size_t)p->D * sizeof(double), inputs, (size_t)ld * sizeof(double),
                             (size_t)p->D * sizeof(double), (size_t)p->n, hipMemcpyHostToDevice,
                             p->stream));
    GPR_HIP(hipStreamSynchronize(p->stream));
    p->have_inputs = true;
  });
}

int gprhip_set_targets(gprhip_problem* p, const double* targets) {
  return guarded([&] {
    if (!p || !targets) {
      set_error("gprhip_set_targets: invalid arguments");
      throw HipFail{ST_BAD_ARG};
    }
    GPR_HIP(hipSetDevice(p->device));
    GPR_HIP(hipMemcpyAsync(p->y, targets, (size_t)p->n * sizeof(double), hipMemcpyHostToDevice, p->stream));
    GPR_HIP(hipStreamSynchronize(p->stream));
    p->have_targets = true;
  });
}

int gprhip_set_inputs_device(gprhip_problem* p, const double* d_inputs) {
  return guarded([&] {
    if (!p || !d_inputs) {
      set_error("gprhip_set_inputs_device: invalid arguments");
      throw HipFail{ST_BAD_ARG};
    }
    GPR_HIP(hipSetDevice(p->device));
    GPR_HIP(hipMemcpyAsync(p->X, d_inputs, (size_t)p->n * p->D * sizeof(double), hipMemcpyDeviceToDevice,
                           p->stream));
    GPR_HIP(hipStreamSynchronize(p->stream));
    p->have_inputs = true;
  });
}

int gprhip_set_targets_device(gprhip_problem* p, const double* d_targets) {
  return guarded([&] {
    if (!p || !d_targets) {
      set_error("gprhip_set_targets_device: invalid arguments");
      throw HipFail{ST_BAD_ARG};
    }
    GPR_HIP(hipSetDevice(p->device));
    GPR_HIP(hipMemcpyAsync(p->y, d_targets, (size_t)p->n * sizeof(double), hipMemcpyDeviceToDevice,
                           p->stream));
    GPR_HIP(hipStreamSynchronize(p->stream));
    p->have_targets = true;
  });
}

int64_t gprhip_n_hypers(const gprhip_problem* p, int flags) {
  if (!p) return 0;
  if (p->kind == GPRHIP_COV_SE_ISO) return 2 + (int64_t)p->d * p->m;
  return 1 + (int64_t)p->d * p->m + ((flags & 1) ? (int64_t)p->D * p->d : 0) + ((flags & 2) ? p->m : 0) +
         ((flags & 4) ? (int64_t)p->d * p->m : 0);
}

int gprhip_memory_plan(int cov_kind, int precision, int64_t n, int D, int d, int m, int64_t chunk_rows,
                       gprhip_memory_plan_t* plan) {
  return guarded([&] {
    if (!plan || n < 1 || D < 1 || d < 1 || m < 1 || (cov_kind != GPRHIP_COV_SE_ISO && cov_kind != GPRHIP_COV_SE_FAT) ||
        (precision != GPRHIP_F64 && precision != GPRHIP_F32_BULK)) {
      set_error("gprhip_memory_plan: invalid arguments");
      throw HipFail{ST_BAD_ARG};
    }
    memory_plan(cov_kind, precision, n, D, d, m, chunk_rows, plan);
  });
}

int64_t gprhip_exchange_len(int cov_kind, int D, int d, int m, int which) {
  if (m < 1 || d < 1 || D < 1 || (which != 1 && which != 2) || (cov_kind != GPRHIP_COV_SE_ISO && cov_kind != GPRHIP_COV_SE_FAT))
    return 0;
  const int mp = (int)round_up(m, TILE);
  if (which == 1) return packed_upper_len(mp) + mp + A1_TAIL;
  const bool fat = cov_kind == GPRHIP_COV_SE_FAT;
  const int64_t dbig = fat ? D : 0, col_rows = d + 1 + dbig + (fat ? d : 0);  // gprhip_problem::col_rows / dbig
  return packed_upper_len(mp) + col_rows * mp + dbig * d + A2_TAIL;
}
int64_t gprhip_exchange_offset(int m, int r, int c) {
  const int mp = (int)round_up(m, TILE);
  if (m < 1 || r < 0 || c < 0 || r >= mp || c >= mp || r / TILE > c / TILE) return -1;
  return packed_upper_off(r, c);
}
int64_t gprhip_ar1_len(const gprhip_problem* p) { return p ? gprhip_exchange_len(p->kind, p->D, p->d, p->m, 1) : 0; }
int64_t gprhip_ar2_len(const gprhip_problem* p) { return p ? gprhip_exchange_len(p->kind, p->D, p->d, p->m, 2) : 0; }

int gprhip_eval_pass1(gprhip_problem* p, const gprhip_hypers* h, int want_grad, int64_t n_total,
                      double* d_ar1) {
  return guarded([&] {
    if (!p || !d_ar1) {
      set_error("gprhip_eval_pass1: NULL argument");
      throw HipFail{ST_BAD_ARG};
    }
    if (p->f32) do_pass1<float>(p, h, want_grad, n_total, d_ar1);
    else do_pass1<double>(p, h, want_grad, n_total, d_ar1);
  }, p);
}

int gprhip_eval_pass2(gprhip_problem* p, const double* d_ar1, double* d_ar2) {
  return guarded([&] {
    if (!p || !d_ar1 || !d_ar2) {
      set_error("gprhip_eval_pass2: NULL argument");
      throw HipFail{ST_BAD_ARG};
    }
    // keep the reduced exchange-1 buffer where finish() reads its scalar tail
    if (d_ar1 != p->ar1)
      GPR_HIP(hipMemcpyAsync(p->ar1, d_ar1, (size_t)gprhip_ar1_len(p) * sizeof(double),
                             hipMemcpyDeviceToDevice, p->stream));
    if (p->f32) do_pass2<float>(p, p->ar1, d_ar2);
    else do_pass2<double>(p, p->ar1, d_ar2);
  }, p);
}

int gprhip_eval_finish(gprhip_problem* p, const double* d_ar2, gprhip_result* res, double* grad,
                       double* coeffs) {
  return guarded([&] {
    if (!p || !d_ar2 || !res || (p->want_grad && !grad)) {
      set_error("gprhip_eval_finish: NULL argument");
      throw HipFail{ST_BAD_ARG};
    }
    do_finish_enqueue(p, d_ar2);
    do_finish_collect(p, res, grad, coeffs);
  }, p);
}

int gprhip_eval(gprhip_problem* p, const gprhip_hypers* h, int want_grad, gprhip_result* res,
                double* grad, double* coeffs) {
  return guarded([&] {
    if (!p || !res || (want_grad && !grad)) {
      set_error("gprhip_eval: NULL argument");
      throw HipFail{ST_BAD_ARG};
    }
    if (p->f32) {
      do_pass1<float>(p, h, want_grad, p->n, p->ar1);
      do_pass2<float>(p, p->ar1, p->ar2);
    } else {
      do_pass1<double>(p, h, want_grad, p->n, p->ar1);
      do_pass2<double>(p, p->ar1, p->ar2);
    }
    do_finish_enqueue(p, p->ar2);
    do_finish_collect(p, res, grad, coeffs);
  }, p);
}

int gprhip_predict(gprhip_problem* p, const double* test_inputs, int64_t ld, int64_t nt, int predictive,
                   double* means, double* variances) {
  return guarded([&] {
    if (!p || !test_inputs || nt < 1 || ld < p->D) {
      set_error("gprhip_predict: invalid arguments");
      throw HipFail{ST_BAD_ARG};
    }
    if (p->f32) do_predict<float>(p, test_inputs, ld, nt, predictive, means, variances);
    else do_predict<double>(p, test_inputs, ld, nt, predictive, means, variances);
  });
}

int gprhip_train_stats(gprhip_problem* p, double* means, double* sums) {
  return guarded([&] {
    if (!p || !sums) {
      set_error("gprhip_train_stats: invalid arguments");
      throw HipFail{ST_BAD_ARG};
    }
    if (p->f32) do_train_stats<float>(p, means, sums);
    else do_train_stats<double>(p, means, sums);
  });
}

int gprhip_covariances(gprhip_problem* p, const double* test_inputs, int64_t ld, int64_t nt, int kind,
                       int predictive, double* cov) {
  return guarded([&] {
    if (!p || !test_inputs || !cov || nt < 1 || ld < p->D || (kind != 0 && kind != 1) || nt > (1 << 20)) {
      set_error("gprhip_covariances: invalid arguments");
      throw HipFail{ST_BAD_ARG};
    }
    do_covariances(p, test_inputs, ld, nt, kind, predictive, cov);
  });
}

int gprhip_cov_samples(gprhip_problem* p, const double* cov, int64_t ld, int64_t nt, double add_diag,
                       double jitter, const double* means, const double* z, int64_t ns, double* samples) {
  return guarded([&] {
    if (!p || !cov || !means || !z || !samples || nt < 1 || ns < 1 || ld < nt || nt > (1 << 20)) {
      set_error("gprhip_cov_samples: invalid arguments");
      throw HipFail{ST_BAD_ARG};
    }
    do_cov_samples(p, cov, ld, nt, add_diag, jitter, means, z, ns, samples);
  });
}

int gprhip_co_variance_coeffs(gprhip_problem* p, double* chol_km, double* r_mat) {
  return guarded([&] {
    if (!p) {
      set_error("gprhip_co_variance_coeffs: invalid arguments");
      throw HipFail{ST_BAD_ARG};
    }
    do_co_variance_coeffs(p, chol_km, r_mat);
  });
}

int gprhip_load_predictor(gprhip_problem* p, const gprhip_hypers* h, const double* coeffs, const double* chol_km,
                          const double* r_mat) {
  return guarded([&] {
    if (!p || !h || (chol_km == nullptr) != (r_mat == nullptr) || (!coeffs && !chol_km)) {
      set_error("gprhip_load_predictor: invalid arguments");
      throw HipFail{ST_BAD_ARG};
    }
    do_load_predictor(p, h, coeffs, chol_km, r_mat);
  });
}

int gprhip_condition(gprhip_problem* p, double* cond_km, double* coeff_error_bound) {
  return guarded([&] {
    if (!p) {
      set_error("gprhip_condition: NULL problem");
      throw HipFail{ST_BAD_ARG};
    }
    need_model(p, "gprhip_condition");
    const double c = condition_km(p);
    if (cond_km) *cond_km = c;
    if (coeff_error_bound) *coeff_error_bound = c * (p->f32 ? 5.9604644775390625e-8 : 1.1102230246251565e-16);
  });
}

int gprhip_sync(gprhip_problem* p) {
  return guarded([&] {
    if (!p) {
      set_error("gprhip_sync: NULL problem");
      throw HipFail{ST_BAD_ARG};
    }
    GPR_HIP(hipSetDevice(p->device));
    GPR_HIP(hipStreamSynchronize(p->stream));
  });
}

void* gprhip_stream(gprhip_problem* p) { return p ? (void*)p->stream : nullptr; }

int gprhip_set_timing(gprhip_problem* p, int level) {
  return guarded([&] {
    if (!p || level < 0 || level > 2) {
      set_error("gprhip_set_timing: invalid arguments");
      throw HipFail{ST_BAD_ARG};
    }
    p->timer.kernel = level >= 1;
    p->timer.on = level >= 2;
  });
}

int gprhip_debug_fetch(gprhip_problem* p, const char* name, double* out, int64_t len) {
  return guarded([&] {
    if (!p || !out) {
      set_error("gprhip_debug_fetch: NULL argument");
      throw HipFail{ST_BAD_ARG};
    }
    const double* src = nullptr;
    int64_t avail = 0;
    std::string nm = name ? name : "";
    // per-row vectors are stored chunk by chunk with padding; only nchunks == 1 keeps them contiguous
    if (nm == "r") { src = p->r; avail = p->n; }
    else if (nm == "is") { src = p->is; avail = p->n; }
    else if (nm == "v") { src = p->v; avail = p->n; }
    else if (nm == "w") { src = p->w; avail = p->n; }
    else if (nm == "t") { src = p->tvec; avail = p->m; }
    else if (nm == "w_mat" || nm == "x_rows") {
      // the gradient factors of Trained.prepare_hyper (lib/fitc_gp.ml:1192-1207) as the last gradient evaluation left
      // them: W (m x m, symmetric, Fortran) and the first rows of X (Fortran rows x m) -- with "km", "knm_rows" and "v"
      // they let a test contract finite differences of the covariance matrices exactly as Shared.calc_log_evidence does
      // (lib/fitc_gp.ml:1005-1021)
      need_model(p, "gprhip_debug_fetch");
      if (!p->want_grad) {
        set_error("gprhip_debug_fetch: the last evaluation was evidence-only");
        throw HipFail{ST_STATE};
      }
      GPR_HIP(hipSetDevice(p->device));
      const int m = p->m, mp = p->mp;
      if (nm == "w_mat") {
        if (len < (int64_t)m * m) {
          set_error("gprhip_debug_fetch: \"w_mat\" needs m*m doubles");
          throw HipFail{ST_BAD_ARG};
        }
        std::vector<double> h((size_t)mp * mp);
        GPR_HIP(hipMemcpyAsync(h.data(), p->wmat, h.size() * sizeof(double), hipMemcpyDeviceToHost, p->stream));
        GPR_HIP(hipStreamSynchronize(p->stream));
        for (int c = 0; c < m; ++c)
          for (int r = 0; r < m; ++r) out[(size_t)c * m + r] = h[(size_t)r * mp + c];
        return;
      }
      const int64_t rows = len / m;
      if (!p->x_last || p->f32 || rows < 1 || rows > p->rows_of(0)) {
        set_error("gprhip_debug_fetch: \"x_rows\" needs an fp64 problem whose rows fit one chunk, and len = rows*m");
        throw HipFail{ST_BAD_ARG};
      }
      std::vector<double> h((size_t)rows * mp);
      GPR_HIP(hipMemcpyAsync(h.data(), p->x_last, h.size() * sizeof(double), hipMemcpyDeviceToHost, p->stream));
      GPR_HIP(hipStreamSynchronize(p->stream));
      for (int c = 0; c < m; ++c)
        for (int64_t r = 0; r < rows; ++r) out[(size_t)c * rows + r] = h[(size_t)r * mp + c];
      return;
    }
    else if (nm == "km" || nm == "knm_rows") {
      // element-wise pins of the covariance kernels (the counterpart of Test.check_deriv_hyper's matrix checks,
      // lib/fitc_gp.ml:1223-1396): K_m as stored, and the first rows of K_nm rebuilt by the chunk builder
      need_model(p, "gprhip_debug_fetch");
      GPR_HIP(hipSetDevice(p->device));
      const int m = p->m, mp = p->mp;
      if (nm == "km") {
        if (len < (int64_t)m * m) {
          set_error("gprhip_debug_fetch: \"km\" needs m*m doubles");
          throw HipFail{ST_BAD_ARG};
        }
        std::vector<double> h((size_t)mp * mp);
        GPR_HIP(hipMemcpyAsync(h.data(), p->km, h.size() * sizeof(double), hipMemcpyDeviceToHost, p->stream));
        GPR_HIP(hipStreamSynchronize(p->stream));
        for (int c = 0; c < m; ++c)
          for (int r = 0; r < m; ++r) out[(size_t)c * m + r] = (r <= c) ? h[(size_t)r * mp + c] : 0.0;
        return;
      }
      const int64_t rows = len / m;
      if (rows < 1 || rows > p->rows_of(0) || !p->have_inputs) {
        set_error("gprhip_debug_fetch: \"knm_rows\" needs len = rows*m with 1 <= rows <= the first row chunk");
        throw HipFail{ST_BAD_ARG};
      }
      void* const kbuf = (p->x_last == p->bufA) ? p->bufB : p->bufA;  // the chunk buffer that does not hold X
      std::vector<double> h((size_t)rows * mp);
      if (p->f32) {
        cov_chunk<float>(p, 0, static_cast<float*>(kbuf));
        std::vector<float> hf((size_t)rows * mp);
        GPR_HIP(hipMemcpyAsync(hf.data(), kbuf, hf.size() * sizeof(float), hipMemcpyDeviceToHost, p->stream));
        GPR_HIP(hipStreamSynchronize(p->stream));
        for (size_t i = 0; i < hf.size(); ++i) h[i] = hf[i];
      } else {
        cov_chunk<double>(p, 0, static_cast<double*>(kbuf));
        GPR_HIP(hipMemcpyAsync(h.data(), kbuf, h.size() * sizeof(double), hipMemcpyDeviceToHost, p->stream));
        GPR_HIP(hipStreamSynchronize(p->stream));
      }
      for (int c = 0; c < m; ++c)
        for (int64_t r = 0; r < rows; ++r) out[(size_t)c * rows + r] = h[(size_t)r * mp + c];
      return;
    }
    else {
      set_error("gprhip_debug_fetch: unknown name");
      throw HipFail{ST_BAD_ARG};
    }
    if (len > avail) len = avail;
    GPR_HIP(hipSetDevice(p->device));
    GPR_HIP(hipStreamSynchronize(p->stream));
    if (avail == p->m || p->nchunks == 1) {
      GPR_HIP(hipMemcpy(out, src, (size_t)len * sizeof(double), hipMemcpyDeviceToHost));
    } else {
      for (int c = 0; c < p->nchunks; ++c) {
        int64_t lo = (int64_t)c * p->chunk;
        if (lo >= len) break;
        int64_t cnt = std::min<int64_t>(p->rows_of(c), len - lo);
        GPR_HIP(hipMemcpy(out + lo, src + lo, (size_t)cnt * sizeof(double), hipMemcpyDeviceToHost));
      }
    }
  });
}

int gprhip_last_timings(gprhip_problem* p, const char** names, float* ms, int cap) {
  if (!p || !names || !ms) return 0;
  int n = (int)std::min<size_t>(p->tnames.size(), (size_t)cap);
  for (int i = 0; i < n; ++i) {
    names[i] = p->tnames[i].c_str();
    ms[i] = p->tms[i];
  }
  return n;
}

}  // extern "C"
