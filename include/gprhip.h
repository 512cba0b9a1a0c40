/* gprhip -- C ABI of the MI355X-native FITC/SPGP core (libgprhip.so).
 *
 * Drop-in boundary for one path of mmottl/gpr: one evaluation of the FITC log evidence and its
 * full hyper-parameter gradient, i.e. what Gpr.Fitc_gp.Make_deriv(Cov_se_iso.Deriv).FITC /
 * (Cov_se_fat.Deriv) compute in
 *     Deriv.Inducing.calc -> Deriv.Inputs.calc -> Deriv.Model.calc -> Deriv.Trained.calc ->
 *     Trained.calc_log_evidence_sigma2 -> Trained.prepare_hyper -> Trained.calc_log_evidence (per hyper)
 * (reference lib/fitc_gp.ml:881-888, :902-911, :1051-1078, :1158-1207, :1005-1021; driven by
 * multim_dcommon lib/fitc_gp.ml:1612-1636).  The reference has no native stubs of its own for this
 * path (it reaches C only through Lacaml); these entry points are what an OCaml stub layer, or the
 * ctypes host layer in gpr_amd/, binds.  See INTEGRATION.md for the OCaml `external` side.
 *
 * Conventions
 *   - All matrices cross the boundary as raw double pointers in the reference's own layout:
 *     Fortran (column-major) Bigarrays -- inputs D x n (one point per column), inducing d x m,
 *     tproj D x d (lib/interfaces.ml:190-195, bin/ocaml_gpr.ml:196-202).
 *   - Host pointers are borrowed for the duration of the call only.
 *   - Every function returns 0 on success or a GPRHIP_E* code; gprhip_last_error() gives the
 *     message (thread-local).  No C++ exception crosses the boundary.
 *   - A context/problem must not be used from two host threads at once.  The one exception is lifetime: the destroy
 *     calls of a context and of its sharded problems may come from different threads in any order (finalisers of a
 *     garbage-collected host on another domain); that bookkeeping is guarded inside the library.
 */
#ifndef GPRHIP_H
#define GPRHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  GPRHIP_OK = 0,
  GPRHIP_EBADARG = 1,    /* reference: Failure / Invalid_argument on argument checks            */
  GPRHIP_ENOTPOSDEF = 2, /* reference: Lacaml Failure on potrf info > 0                          */
  GPRHIP_EHIP = 3,       /* HIP runtime error                                                    */
  GPRHIP_EOOM = 4,       /* device out of memory                                                 */
  GPRHIP_ESTATE = 5,     /* call sequence violated (e.g. pass2 before pass1)                     */
  GPRHIP_ECOMM = 6,      /* RCCL could not be loaded / initialised, or a collective failed       */
  GPRHIP_EPRECISION = 7  /* fp32-bulk mean coefficients refused: K_m too ill-conditioned for them */
};

enum { GPRHIP_COV_SE_ISO = 0, GPRHIP_COV_SE_FAT = 1 }; /* lib/cov_se_iso.ml, lib/cov_se_fat.ml */

/* Arithmetic of the n x m contractions.  The reference is fp64 only (Lacaml.D, lib/fitc_gp.ml:23).
 *   GPRHIP_F64      : everything fp64 (reference parity).
 *   GPRHIP_F32_BULK : K_nm and every n x m intermediate stored in fp32, n x m x m contractions on the
 *                     fp32 MFMA with split-K partial sums combined in fp64; covariance evaluation,
 *                     row reductions, and all m x m factorisations/solves stay fp64 (BASELINE.json
 *                     config 3).  Agreement with the fp64 reference is ~1e-5 relative. */
enum { GPRHIP_F64 = 0, GPRHIP_F32_BULK = 1 };

typedef struct gprhip_problem gprhip_problem;

/* Number of visible HIP devices (does not initialise a device). */
int gprhip_device_count(int* count);

/* Device memory one shard of a problem holds, in bytes (pure arithmetic, no device): what gprhip_problem_create allocates
 * plus the resident V = K_nm U^-1 store the first evaluation adds.  gprhip_problem_create compares `total` with the free
 * memory of its device BEFORE allocating anything and returns GPRHIP_EOOM with these figures in the message when the shard
 * cannot fit (nothing is spilled to the host); GPRHIP_VERBOSE=1 prints the plan of every problem created.
 * Replaces: nothing -- the reference allocates Bigarrays as it goes (lib/fitc_gp.ml:105-229) and dies in caml_ba_alloc. */
typedef struct {
  int64_t total;            /* sum of the fields below except k_store_optional */
  int64_t v_store;          /* V = K_nm U^-1 of all rows of the shard (n rounded up to chunks x m rounded up to 128) */
  int64_t chunk_buffers;    /* two row-chunk temporaries (K / Q' / X~ / X) and the GEMM epilogues' per-row partials */
  int64_t slices;           /* split-K partial sums of the SYRK-shaped launches */
  int64_t inputs;           /* training inputs, targets, projected inputs (Cov_se_fat) */
  int64_t row_vectors;      /* r, 1/s, y/s, w, v (, E row sums) */
  int64_t mxm;              /* the m x m matrices: K_m, U, U^-1, B~, R~^-1, B~^-1, W~, W, R^-1, scratch */
  int64_t rest;             /* exchange buffers, gradient / trace partials, result block */
  int64_t k_store_optional; /* Cov_se_fat: a second n x m matrix (K_nm kept for the gradient pass), taken only while it
                               leaves 4 GB free and stays below 40 % of the device; not in `total` */
  int64_t chunk_rows;       /* rows per chunk the problem will use */
  int64_t kslices;          /* split-K slices reserved */
} gprhip_memory_plan_t;
int gprhip_memory_plan(int cov_kind, int precision, int64_t n, int D, int d, int m, int64_t chunk_rows,
                       gprhip_memory_plan_t* plan);

/* Create the device-resident state for one shard of a FITC problem on HIP device `device`:
 *   n  training points held by this shard, D input dimension, d kernel-space dimension
 *   (d == D for Cov_se_iso; Cov_se_fat: d = Params.d, the tproj target dimension), m inducing points.
 *   chunk_rows: rows of K_nm processed per streaming step (0 = library default).
 * Replaces: the Bigarray allocations spread over Eval_inputs / Eval_model (lib/fitc_gp.ml:105-229). */
int gprhip_problem_create(int device, int cov_kind, int64_t n, int D, int d, int m, int64_t chunk_rows,
                          gprhip_problem** out);
int gprhip_problem_create_ex(int device, int cov_kind, int precision, int64_t n, int D, int d, int m,
                             int64_t chunk_rows, gprhip_problem** out);
void gprhip_problem_destroy(gprhip_problem* p);

/* Training inputs (Fortran D x n, leading dimension ld >= D) and targets (n): copied to the device.
 * Replaces: Spec.Inputs.t / targets arguments of Deriv.Inputs.calc and Deriv.Trained.calc. */
int gprhip_set_inputs(gprhip_problem* p, const double* inputs, int64_t ld);
int gprhip_set_targets(gprhip_problem* p, const double* targets);
/* Same, from device pointers (inputs already resident in HBM; point-major [n][D], contiguous). */
int gprhip_set_inputs_device(gprhip_problem* p, const double* d_inputs);
int gprhip_set_targets_device(gprhip_problem* p, const double* d_targets);

/* Hyper-parameters of one evaluation.
 *   log_ell    : Cov_se_iso.Params.log_ell (ignored for Cov_se_fat)          lib/cov_se_iso.ml:23-25
 *   log_sf2    : Params.log_sf2
 *   sigma2     : noise variance (>= 0, else GPRHIP_EBADARG: lib/fitc_gp.ml:148-149)
 *   inducing   : Fortran d x m inducing points (Spec.Inducing.t)
 *   tproj      : Fortran D x d projection (Cov_se_fat.Params.tproj) or NULL
 *   variational: 0 = Make_FITC_deriv (Standard), 1 = Make_variational_FITC_deriv  lib/fitc_gp.ml:262-263
 *   model_only : 1 = evidence/gradient of the model without targets (Deriv.Model.*), 0 = Trained.*
 *   jitter     : Utils.cholesky_jitter (lib/utils.ml:35); pass 1e-6 for the reference behaviour
 *   log_hetero_skedasticity : Cov_se_fat.Params.log_hetero_skedasticity (m entries) or NULL:
 *                exp() of it is added to diag(K_m)                             lib/cov_se_fat.ml:136-142
 *   log_multiscales_m05 : Cov_se_fat.Params.log_multiscales_m05 (Fortran d x m) or NULL: per inducing point
 *                and dimension a length scale exp(.)+0.5                        lib/cov_se_fat.ml:66-69, :115-134
 *   reuse_v    : 1 = only sigma2 (and targets) changed since the previous evaluation on this problem:
 *                K_nm, V = K_nm U^-1 and r are reused instead of recomputed -- Model.update_sigma2,
 *                lib/fitc_gp.ml:234-236, :1083-1090.  The caller guarantees kernel parameters and inducing
 *                points are unchanged (they are still passed and uploaded). */
typedef struct {
  double log_ell;
  double log_sf2;
  double sigma2;
  const double* inducing;
  const double* tproj;
  int variational;
  int model_only;
  double jitter;
  const double* log_hetero_skedasticity;
  const double* log_multiscales_m05;
  int reuse_v;
} gprhip_hypers;

/* Results.  Gradient order is the reference's Hyper.get_all order:
 *   Cov_se_iso: [Log_ell; Log_sf2; Inducing_hyper{ind=1,dim=1..d}; {ind=2,..}; ...]   lib/cov_se_iso.ml:188-202
 *   Cov_se_fat: [Log_sf2; Inducing_hyper (ind-major); Proj{big_dim,small_dim} (big-major);
 *                Log_hetero_skedasticity 1..m; Log_multiscale_m05 (ind-major)]  lib/cov_se_fat.ml:290-342
 * l1 = Model.calc_log_evidence, l = Trained.calc_log_evidence, dl_dsigma2 = calc_log_evidence_sigma2. */
typedef struct {
  double l1;
  double l2;
  double l;
  double dl_dsigma2;
  int64_t n_hypers;
} gprhip_result;

/* flags: bit 0 = tproj given, bit 1 = log_hetero_skedasticity given, bit 2 = log_multiscales_m05 given */
int64_t gprhip_n_hypers(const gprhip_problem* p, int flags);

/* One complete evaluation on one device (shard == whole problem).
 *   want_grad = 0: log evidence only (multim_f, lib/fitc_gp.ml:1601-1610)
 *   want_grad = 1: + dl_dsigma2 and grad[n_hypers] (multim_dcommon, lib/fitc_gp.ml:1612-1636)
 *   coeffs (m, may be NULL): Trained.calc_mean_coeffs (lib/fitc_gp.ml:294). */
int gprhip_eval(gprhip_problem* p, const gprhip_hypers* h, int want_grad, gprhip_result* res,
                double* grad, double* coeffs);

/* Staged form for row-sharded evaluation across devices (one process per device).  Between the
 * stages the caller sums the exchange buffers over all shards (RCCL all-reduce on the same HIP
 * stream or after a stream sync -- the buffers are plain device memory owned by the caller):
 *     gprhip_eval_pass1(p, h, want_grad, ar1)      ar1: gprhip_ar1_len(p) doubles
 *     all-reduce(sum, ar1)
 *     gprhip_eval_pass2(p, ar1, ar2)               ar2: gprhip_ar2_len(p) doubles
 *     all-reduce(sum, ar2)
 *     gprhip_eval_finish(p, ar2, res, grad, coeffs)
 * n used in the n*log(2*pi) term is n_total given here (sum over shards). */
int64_t gprhip_ar1_len(const gprhip_problem* p);
int64_t gprhip_ar2_len(const gprhip_problem* p);
/* The same lengths from the problem's dimensions alone (no device, no handle; which = 1 | 2), and the position of entry
 * (r, c) of the symmetric m x m part that heads both buffers -- stored as its upper 128 x 128 tiles only, tile (bm <= bn)
 * at ((bn (bn + 1) / 2 + bm) * 128 * 128, row-major inside (r, c < m rounded up to 128; the tile of (r, c) must be on or
 * above the diagonal, else -1).  A host that owns the collective sizes and fills its buffers with these. */
int64_t gprhip_exchange_len(int cov_kind, int D, int d, int m, int which);
int64_t gprhip_exchange_offset(int m, int r, int c);
int gprhip_eval_pass1(gprhip_problem* p, const gprhip_hypers* h, int want_grad, int64_t n_total,
                      double* d_ar1);
int gprhip_eval_pass2(gprhip_problem* p, const double* d_ar1, double* d_ar2);
int gprhip_eval_finish(gprhip_problem* p, const double* d_ar2, gprhip_result* res, double* grad,
                       double* coeffs);
/* Block the host until all work queued by the stage calls has finished. */
int gprhip_sync(gprhip_problem* p);
/* The HIP stream (hipStream_t) the problem enqueues on, for callers that order other work after it. */
void* gprhip_stream(gprhip_problem* p);

/* ---- Single-process, multi-device evaluation ---------------------------------------------------------------------
 * The reference's host is ONE process (bin/ocaml_gpr.ml:176-177, :340-342: one functor application, one optimiser
 * loop), so the way it reaches the GPUs of a node is a context that owns one shard per device and does the exchange
 * steps itself -- the calls above with "the caller does the all-reduce" serve one-process-per-GPU hosts
 * (gpr_amd/dist.py under torch.distributed).
 *
 *   gprhip_ctx_create(devices, ndev)        the devices of the node to use; loads RCCL (dlopen, so that single-device
 *                                           use of the library has no RCCL dependency) and creates one communicator
 *                                           per device (ncclCommInitAll) when ndev > 1
 *   gprhip_sharded_create(ctx, ...)         one gprhip_problem per device; shard i owns the contiguous training rows
 *                                           [lo_i, hi_i) (sizes differ by at most one, as gpr_amd.dist.shard_rows)
 *   gprhip_sharded_set_inputs / _targets    the whole D x n / n host arrays; every shard copies its own rows
 *   gprhip_sharded_eval                     pass 1 on every device (one host thread per device enqueues on that
 *                                           device's stream), grouped ncclAllReduce(sum, fp64) of the packed exchange
 *                                           buffers on those same streams -- no host synchronisation, no event hop --
 *                                           pass 2, second all-reduce (gradient evaluations only), finish on the first
 *                                           device.  Results are those of gprhip_eval on the unsharded problem up to the
 *                                           summation order of the exchange buffers; with ndev == 1 they are bit-identical.
 * Validation mode: if `devices` names ONE device ndev > 1 times, the shards share that device and the exchange step is
 * a fixed-order device-local sum instead of RCCL (which refuses duplicate devices) -- the ndev-way partition and all of
 * its bookkeeping can then be exercised on a one-GPU box.  Mixed lists (some devices repeated) are refused.
 * GPRHIP_CTX_RCCL=1 in the environment at creation makes a one-device context go through RCCL as well (a one-rank
 * communicator): exercises the dlopen / ncclCommInitAll / ncclAllReduce path on a one-GPU box.
 * GPRHIP_RCCL_LIB names the shared object to load (default: librccl.so.1, then librccl.so, then /opt/rocm/lib). */
typedef struct gprhip_ctx gprhip_ctx;
typedef struct gprhip_sharded gprhip_sharded;

enum { GPRHIP_COMM_NONE = 0, GPRHIP_COMM_RCCL = 1, GPRHIP_COMM_SAME_DEVICE = 2 };

/* The row partition itself (pure arithmetic, no device): shard idx of ndev owns rows [*row_lo, *row_hi) of n. */
int gprhip_shard_rows(int64_t n, int ndev, int idx, int64_t* row_lo, int64_t* row_hi);

int gprhip_ctx_create(const int* devices, int ndev, gprhip_ctx** out);
/* May be called while sharded problems of the context are still alive (a garbage-collected host finalises handles in
 * any order): the context is then released together with the last of them. */
void gprhip_ctx_destroy(gprhip_ctx* ctx);
int gprhip_ctx_ndev(const gprhip_ctx* ctx);
int gprhip_ctx_comm_mode(const gprhip_ctx* ctx); /* GPRHIP_COMM_* */

/* Arguments as gprhip_problem_create_ex, n = training points of the WHOLE problem (>= ndev). */
int gprhip_sharded_create(gprhip_ctx* ctx, int cov_kind, int precision, int64_t n, int D, int d, int m,
                          int64_t chunk_rows, gprhip_sharded** out);
void gprhip_sharded_destroy(gprhip_sharded* sp);
/* Shard idx (0 <= idx < ndev): its device, its row range [*row_lo, *row_hi) of the whole problem (any pointer may be
 * NULL), and its device problem -- the m x m model state is replicated, so gprhip_predict, gprhip_covariances,
 * gprhip_co_variance_coeffs, ... work on any shard's problem after an evaluation; gprhip_train_stats and the per-row
 * names of gprhip_debug_fetch cover that shard's rows. */
int gprhip_sharded_shard(const gprhip_sharded* sp, int idx, int* device, int64_t* row_lo, int64_t* row_hi);
gprhip_problem* gprhip_sharded_problem(gprhip_sharded* sp, int idx);
int gprhip_sharded_set_inputs(gprhip_sharded* sp, const double* inputs, int64_t ld); /* Fortran D x n, host */
int gprhip_sharded_set_targets(gprhip_sharded* sp, const double* targets);           /* n, host          */
/* As gprhip_eval. */
int gprhip_sharded_eval(gprhip_sharded* sp, const gprhip_hypers* h, int want_grad, gprhip_result* res, double* grad,
                        double* coeffs);
/* Posterior prediction with all devices (gprhip_predict on every shard's problem for its share of the test points --
 * the model state is replicated, the test points are split like the training rows): means / variances as there. */
int gprhip_sharded_predict(gprhip_sharded* sp, const double* test_inputs, int64_t ld, int64_t nt, int predictive,
                           double* means, double* variances);
/* Training-set statistics over all shards (gprhip_train_stats per shard, combined sum / sum / max / sum): means (n,
 * whole problem, may be NULL) and sums[4]. */
int gprhip_sharded_train_stats(gprhip_sharded* sp, double* means, double* sums);
/* Exchange steps of the last evaluation: their count (1 evidence-only, 2 gradient; 0 with one device and no forced
 * RCCL), bytes per device of each, and -- after gprhip_sharded_set_timing(sp, 1) -- their milliseconds on the first
 * shard's stream (HIP events; includes waiting for the slowest shard). */
int gprhip_sharded_comm_stats(const gprhip_sharded* sp, int* collectives, int64_t bytes[2], float ms[2]);
int gprhip_sharded_set_timing(gprhip_sharded* sp, int level);

/* Posterior prediction at test points with the model state left by the last evaluation on `p`
 * (kernel, inducing points, U = chol_km, R = r_mat, mean coefficients):
 *   means[i]     = K_tm[i,:] . coeffs                          Means.calc     lib/fitc_gp.ml:418-425
 *   variances[i] = k_ii - |K_tm U^-1|_i^2 + |K_tm R^-1|_i^2    Variances.calc lib/fitc_gp.ml:498-518
 *                  (+ sigma2 when predictive != 0: Variances.get ?predictive, :520-529)
 * test_inputs: Fortran D x nt (ld >= D), host.  means / variances: nt doubles, host; either may be NULL.
 * The last evaluation must have had targets (model_only = 0) for the means to be meaningful. */
int gprhip_predict(gprhip_problem* p, const double* test_inputs, int64_t ld, int64_t nt, int predictive,
                   double* means, double* variances);

/* Training-set residual statistics with the model state left by the last evaluation (which must have had
 * targets): means[i] = K_nm[i,:] . coeffs over the resident training inputs (Trained.calc_means,
 * lib/fitc_gp.ml:296-297; host, n doubles, may be NULL) and
 *   sums[0] = sum (y-mean)^2   sums[1] = sum |y-mean|   sums[2] = max |y-mean|   sums[3] = sum y^2
 * from which Stats.calc (lib/fitc_gp.ml:353-373) derives sse/mse/rmse/smse/msll/mad/maxad.  For a row shard
 * the sums are this shard's; combine with sum/sum/max/sum. */
int gprhip_train_stats(gprhip_problem* p, double* means, double* sums);

/* Posterior covariance matrix between nt test points (fp64 in both precision modes):
 *   kind 0: FITC_covariances.calc  K_tt - V_t V_t^T + Q_t Q_t^T                    lib/fitc_gp.ml:585-599
 *   kind 1: FIC_covariances.calc   Q_t Q_t^T + diag(k_tt - rowsum(K_tm .^ 2))      lib/fitc_gp.ml:617-627
 * with V_t = K_tm U^-1, Q_t = K_tm R^-1.  predictive != 0 adds sigma2 to the diagonal (Common_covariances.get,
 * :549-559).  test_inputs: Fortran D x nt (ld >= D), host.  cov: nt x nt, ld = nt, host; the full symmetric
 * matrix is written (the reference defines the upper triangle only). */
int gprhip_covariances(gprhip_problem* p, const double* test_inputs, int64_t ld, int64_t nt, int kind,
                       int predictive, double* cov);

/* Common_cov_sampler.calc + samples (lib/fitc_gp.ml:656-697): factor chol(cov + (add_diag + jitter) I) (upper
 * triangle of the Fortran nt x nt `cov`, ld >= nt, is read) and return samples[:, j] = means + chol^T z[:, j]
 * for the ns columns of z (Fortran nt x ns standard normal draws supplied by the caller; samples likewise).
 * add_diag = sigma2 for ?predictive = true, else 0; jitter = Utils.cholesky_jitter.  Uses only the device
 * and stream of `p`.  GPRHIP_ENOTPOSDEF if the factorisation fails. */
int gprhip_cov_samples(gprhip_problem* p, const double* cov, int64_t ld, int64_t nt, double add_diag,
                       double jitter, const double* means, const double* z, int64_t ns, double* samples);

/* Model.calc_co_variance_coeffs (lib/fitc_gp.ml:240): the pair (chol_km, r_mat) of the last evaluation, each a
 * Fortran m x m upper-triangular factor (zeros below the diagonal), host; either may be NULL.  Together with the
 * kernel parameters, the inducing points and the mean coefficients this is what bin/ocaml_gpr.ml:207-232 stores
 * in its model file. */
int gprhip_co_variance_coeffs(gprhip_problem* p, double* chol_km, double* r_mat);

/* Install the predictor state of a saved model without evaluating anything -- Mean_predictor.calc
 * (lib/fitc_gp.ml:386-391) + Inducing.calc + Co_variance_predictor.calc (:446-447), the `test` flow of
 * bin/ocaml_gpr.ml:373-413.  h: kernel parameters, inducing points, sigma2 (as for gprhip_eval); coeffs: m mean
 * coefficients (NULL: variances/covariances only); chol_km, r_mat: as returned by gprhip_co_variance_coeffs
 * (both NULL: means only -- variance/covariance calls then return GPRHIP_ESTATE).
 * Afterwards gprhip_predict / gprhip_covariances work on `p`; training inputs need not have been set
 * (create the problem with n = the largest test batch you intend to pass). */
int gprhip_load_predictor(gprhip_problem* p, const gprhip_hypers* h, const double* coeffs, const double* chol_km,
                          const double* r_mat);

/* Conditioning of the inducing covariance of the current model state: *cond_km = an estimate (power iteration on
 * U^T U and U^-1 U^-T, a lower bound within a small factor) of the 2-norm condition number of K_m + jitter, and
 * *coeff_error_bound = cond_km * unit roundoff of the n x m operands (2^-24 for GPRHIP_F32_BULK, 2^-53 for GPRHIP_F64):
 * the relative accuracy the mean coefficients (Trained.calc_mean_coeffs, lib/fitc_gp.ml:294, :288-292) can be trusted
 * to.  Log evidence and gradient do not carry that factor.  Either pointer may be NULL.
 * GPRHIP_F32_BULK problems enforce it: when the bound exceeds GPRHIP_F32_COEFF_TOL (environment, read at problem
 * creation; default 0.25, i.e. cond ~ 4e6 -- the measured coefficient error is 1/17 .. 1/3600 of this worst-case bound,
 * <= ~1.5e-2 at the threshold; 0 = never refuse) gprhip_predict (means) and gprhip_train_stats return GPRHIP_EPRECISION
 * instead of results computed from such coefficients. */
int gprhip_condition(gprhip_problem* p, double* cond_km, double* coeff_error_bound);

/* Intermediates of the last evaluation, for parity tests (copied to host; sizes in doubles):
 *   "r" n, "is" n, "v" n, "w" n, "t" m;
 *   "km"  m*m : K_m as Inducing.calc_upper leaves it (lib/cov_se_iso.ml:56-87, lib/cov_se_fat.ml:110-142, without jitter
 *               or heteroskedastic noise), Fortran m x m, upper triangle valid, zeros below;
 *   "knm_rows" rows*m (rows = len / m <= the first row chunk): the first rows of K_nm recomputed with the kernel of the
 *               last evaluation (Inputs.calc_cross, lib/cov_se_iso.ml:128-159, lib/cov_se_fat.ml:224-256), Fortran
 *               rows x m (column-major, leading dimension rows); fp32-bulk problems return the rounded stored values;
 *   "w_mat" m*m : W of Trained.prepare_hyper (lib/fitc_gp.ml:1196-1203) after a gradient evaluation, Fortran m x m, symmetric;
 *   "x_rows" rows*m : the first rows of X (lib/fitc_gp.ml:1204-1206), Fortran rows x m -- fp64 problems whose training
 *               points fit one row chunk only (the chunk buffers are reused otherwise).
 * Returns GPRHIP_EBADARG for an unknown name. */
int gprhip_debug_fetch(gprhip_problem* p, const char* name, double* out, int64_t len);

/* Timing of evaluations with HIP events on the problem's own stream.  level 0: none (default; GPRHIP_TIMING in the
 * environment sets the initial level); 1: one event pair around the dominant kernel alone (the pass-1 SYRK launch over
 * the shard's training points, reported as "kernel_p1_syrk_B"); 2: also a pair around every stage of the evaluation. */
int gprhip_set_timing(gprhip_problem* p, int level);

/* Timings of the last evaluation (milliseconds): fills up to `cap` entries of names/ms; returns the count. */
int gprhip_last_timings(gprhip_problem* p, const char** names, float* ms, int cap);

const char* gprhip_last_error(void);
const char* gprhip_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GPRHIP_H */
