/* TEST INFRASTRUCTURE -- not part of the product.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this.
 *
 * C restatement of the reference's FITC log evidence + hyper-gradient evaluation for Cov_se_iso, following the
 * reference's own operation sequence call for call (OCaml-GPR 1.5.2, fp64, Fortran layout):
 *   covariances   lib/cov_se_iso.ml:56-87 (K_m upper), :128-159 (K_nm, direct-difference loops), :247-327 (derivatives)
 *   engine        lib/fitc_gp.ml:53-57 (chol K_m), :222-229 (V, r), :151-220 (stacked Householder QR, l1),
 *                 :279-292 / :1158-1181 (trained: l2, coefficients, u, w, v), :1037-1078 (K_m^-1, B^-1, T, q_diag),
 *                 :1092-1119 (v1, dl/dsigma2), :931-939 (U_mat, S), :1192-1207 (W, X), :943-1021 (per-hyper traces)
 *   helpers       lib/utils.ml:35 (jitter), :95-101 (log_det), :110-113 (ichol), :196-220 (sparse-row trace)
 * The reference reaches BLAS/LAPACK through Lacaml; here the same routines (dpotrf, dtrsm, dgeqrf, dorgqr, dpotri,
 * dsyrk, dgemv, dtrsv) are taken from whatever LAPACK shared object the caller names (scipy's bundled OpenBLAS in this
 * image), and the reference's scalar OCaml loops are OpenMP loops.  It is (a) a second oracle, checked against
 * oracle/fitc_oracle.py in tests/test_oracle.py, and (b) the CPU baseline bench.py times on the host cores.
 * Parity status: unpinned by reference fixtures (the reference ships none and cannot be built here) -- see DESIGN.md 6.
 */
#include <dlfcn.h>
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef int bint; /* LP64 LAPACK */
typedef void (*potrf_t)(const char*, const bint*, double*, const bint*, bint*);
typedef void (*potri_t)(const char*, const bint*, double*, const bint*, bint*);
typedef void (*trsm_t)(const char*, const char*, const char*, const char*, const bint*, const bint*, const double*,
                       const double*, const bint*, double*, const bint*);
typedef void (*geqrf_t)(const bint*, const bint*, double*, const bint*, double*, double*, const bint*, bint*);
typedef void (*orgqr_t)(const bint*, const bint*, const bint*, double*, const bint*, const double*, double*,
                        const bint*, bint*);
typedef void (*syrk_t)(const char*, const char*, const bint*, const bint*, const double*, const double*, const bint*,
                       const double*, double*, const bint*);
typedef void (*gemv_t)(const char*, const bint*, const bint*, const double*, const double*, const bint*, const double*,
                       const bint*, const double*, double*, const bint*);
typedef void (*trsv_t)(const char*, const char*, const char*, const bint*, const double*, const bint*, double*,
                       const bint*);
typedef void (*setthr_t)(int);
typedef int (*getthr_t)(void);

static struct {
  void* h;
  potrf_t potrf;
  potri_t potri;
  trsm_t trsm;
  geqrf_t geqrf;
  orgqr_t orgqr;
  syrk_t syrk;
  gemv_t gemv;
  trsv_t trsv;
  setthr_t set_threads;
  getthr_t get_threads;
} L;

static void* sym2(void* h, const char* a, const char* b) {
  void* p = dlsym(h, a);
  return p ? p : dlsym(h, b);
}

static int load_lapack(const char* path) {
  if (L.h) return 0;
  L.h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!L.h) return -1;
  L.potrf = (potrf_t)sym2(L.h, "scipy_dpotrf_", "dpotrf_");
  L.potri = (potri_t)sym2(L.h, "scipy_dpotri_", "dpotri_");
  L.trsm = (trsm_t)sym2(L.h, "scipy_dtrsm_", "dtrsm_");
  L.geqrf = (geqrf_t)sym2(L.h, "scipy_dgeqrf_", "dgeqrf_");
  L.orgqr = (orgqr_t)sym2(L.h, "scipy_dorgqr_", "dorgqr_");
  L.syrk = (syrk_t)sym2(L.h, "scipy_dsyrk_", "dsyrk_");
  L.gemv = (gemv_t)sym2(L.h, "scipy_dgemv_", "dgemv_");
  L.trsv = (trsv_t)sym2(L.h, "scipy_dtrsv_", "dtrsv_");
  L.set_threads = (setthr_t)sym2(L.h, "scipy_openblas_set_num_threads", "openblas_set_num_threads");
  L.get_threads = (getthr_t)sym2(L.h, "scipy_openblas_get_num_threads", "openblas_get_num_threads");
  if (!L.potrf || !L.potri || !L.trsm || !L.geqrf || !L.orgqr || !L.syrk || !L.gemv || !L.trsv) return -2;
  return 0;
}

static double now(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return t.tv_sec + 1e-9 * t.tv_nsec;
}

#define IDX(r, c, ld) ((int64_t)(c) * (ld) + (r))
#define LOG_2PI 1.8378770664093454835606594728112 /* lib/utils.ml:39-40 */
#define JITTER 1e-6                                 /* lib/utils.ml:35 */

/* Returns 0, a negative number for set-up failures, or a positive LAPACK info (potrf: not positive definite).
 * X: d x n, Z: d x m (one point per column), y: n.  out = {l1, l2, l, dl/dsigma2}; grad: 2 + d*m entries in the
 * order of Hyper.get_all (lib/cov_se_iso.ml:188-202); coeffs: m; secs[0..5] = {covariances, factor/QR, trained + inverses,
 * U_mat/S/W/X, per-hyper traces, total}, secs[6] = BLAS threads in use, secs[7] = OpenMP threads. */
int fitc_ref_iso(const char* lapack_so, int64_t n, int m, int d, const double* X, const double* y, const double* Z,
                 double log_ell, double log_sf2, double sigma2, int threads, double* out, double* grad, double* coeffs,
                 double* secs) {
  int rc = load_lapack(lapack_so);
  if (rc) return rc;
  if (threads < 1) threads = 1;
  if (L.set_threads) L.set_threads(threads);
  omp_set_num_threads(threads);
  const double inv_ell2 = exp(-2.0 * log_ell), inv_ell2_05 = -0.5 * inv_ell2, sf2 = exp(log_sf2); /* cov_se_iso.ml:41-44 */
  const bint M = m, N = (bint)n, NM = (bint)(n + m), one = 1;
  const double d1 = 1.0, d0 = 0.0, dm1 = -1.0;
  bint info = 0;
  const int64_t nm = n * (int64_t)m, mm = (int64_t)m * m;
  double* km = malloc(mm * 8);
  double* sd_m = malloc(mm * 8);
  double* chol = malloc(mm * 8);
  double* knm = malloc(nm * 8);
  double* sd_nm = malloc(nm * 8);
  double* vmat = malloc(nm * 8);
  double* qmat = malloc((n + m) * (int64_t)m * 8);
  double* umat = malloc(nm * 8);
  double* xmat = malloc(nm * 8);
  double* u1 = malloc(nm * 8);
  double* rmat = malloc(mm * 8);
  double* tmat = malloc(mm * 8);
  double* wmat = malloc(mm * 8);
  double* inv_b = malloc(mm * 8);
  double* vecs = malloc(12 * n * 8);
  double* tau = malloc((size_t)m * 8);
  double* qty = malloc((size_t)m * 8);
  if (!km || !sd_m || !chol || !knm || !sd_nm || !vmat || !qmat || !umat || !xmat || !u1 || !rmat || !tmat || !wmat ||
      !inv_b || !vecs || !tau || !qty)
    return -3;
  double *r_vec = vecs, *is_vec = vecs + n, *sqrt_is = vecs + 2 * n, *y_ = vecs + 3 * n, *u_vec = vecs + 4 * n,
         *w_vec = vecs + 5 * n, *v_vec = vecs + 6 * n, *v1_vec = vecs + 7 * n, *q_diag = vecs + 8 * n,
         *sq_v1 = vecs + 9 * n;
  double t0 = now(), t_start = t0;

  /* ---- Inducing.calc_upper (cov_se_iso.ml:56-87): strict upper from squared differences, diagonal exactly sf2 */
#pragma omp parallel for schedule(dynamic, 16)
  for (int c = 0; c < m; ++c) {
    for (int r = 0; r < c; ++r) {
      double s = 0.0;
      for (int i = 0; i < d; ++i) {
        const double diff = Z[IDX(i, c, d)] - Z[IDX(i, r, d)];
        s += diff * diff;
      }
      sd_m[IDX(r, c, m)] = s;
      km[IDX(r, c, m)] = exp(log_sf2 + inv_ell2_05 * s);
    }
    sd_m[IDX(c, c, m)] = 0.0;
    km[IDX(c, c, m)] = sf2;
    for (int r = c + 1; r < m; ++r) km[IDX(r, c, m)] = sd_m[IDX(r, c, m)] = 0.0; /* (the reference leaves it unset) */
  }
  /* ---- Inputs.calc_cross (cov_se_iso.ml:128-159) */
#pragma omp parallel for schedule(static)
  for (int c = 0; c < m; ++c) {
    double* kc = knm + IDX(0, c, n);
    double* sc = sd_nm + IDX(0, c, n);
    for (int64_t r = 0; r < n; ++r) {
      double s = 0.0;
      for (int i = 0; i < d; ++i) {
        const double diff = X[IDX(i, r, d)] - Z[IDX(i, c, d)];
        s += diff * diff;
      }
      sc[r] = s;
      kc[r] = exp(log_sf2 + inv_ell2_05 * s);
    }
  }
  secs[0] = now() - t0;
  t0 = now();

  /* ---- Inducing.calc_internal (fitc_gp.ml:53-57): chol(K_m + jitter I), log det */
  memcpy(chol, km, mm * 8);
  for (int i = 0; i < m; ++i) chol[IDX(i, i, m)] += JITTER;
  L.potrf("U", &M, chol, &M, &info);
  if (info) return (int)info;
  double log_det_km = 0.0;
  for (int i = m - 1; i >= 0; --i) log_det_km += log(chol[IDX(i, i, m)]);
  log_det_km += log_det_km;
  /* ---- calc_with_kn_diag (fitc_gp.ml:222-229): V = K_nm U^-1, r = k_diag - rowsum(V.^2) */
  memcpy(vmat, knm, nm * 8);
  L.trsm("R", "U", "N", "N", &N, &M, &d1, chol, &M, vmat, &N);
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < n; ++r) {
    double s = 0.0;
    for (int c = 0; c < m; ++c) s += vmat[IDX(r, c, n)] * vmat[IDX(r, c, n)];
    r_vec[r] = sf2 - s;
  }
  /* ---- Common_model.calc_internal (fitc_gp.ml:151-220) */
  double log_det_s = 0.0;
  for (int64_t i = n - 1; i >= 0; --i) {
    const double s = r_vec[i] + sigma2;
    is_vec[i] = 1.0 / s;
    sqrt_is[i] = sqrt(is_vec[i]);
    log_det_s += log(s);
  }
#pragma omp parallel for schedule(static)
  for (int c = 0; c < m; ++c) {
    double* qc = qmat + IDX(0, c, n + m);
    const double* kc = knm + IDX(0, c, n);
    for (int64_t r = 0; r < n; ++r) qc[r] = kc[r] * sqrt_is[r];
    for (int r = 0; r < m; ++r) qc[n + r] = (r <= c) ? chol[IDX(r, c, m)] : 0.0;
  }
  double wq = 0.0;
  bint lwork = -1;
  L.geqrf(&NM, &M, qmat, &NM, tau, &wq, &lwork, &info);
  lwork = (bint)wq;
  double* work = malloc((size_t)lwork * 8);
  L.geqrf(&NM, &M, qmat, &NM, tau, work, &lwork, &info);
  if (info) return -4;
  for (int c = 0; c < m; ++c)
    for (int r = 0; r < m; ++r) rmat[IDX(r, c, m)] = (r <= c) ? qmat[IDX(r, c, n + m)] : 0.0;
  bint lw2 = -1;
  L.orgqr(&NM, &M, &M, qmat, &NM, tau, &wq, &lw2, &info);
  lw2 = (bint)wq;
  if (lw2 > lwork) {
    free(work);
    work = malloc((size_t)lw2 * 8);
  }
  L.orgqr(&NM, &M, &M, qmat, &NM, tau, work, &lw2, &info);
  free(work);
  if (info) return -5;
  double log_det_r = 0.0;
  for (int r = m - 1; r >= 0; --r) { /* sign fix, fitc_gp.ml:183-203 */
    double el = rmat[IDX(r, r, m)];
    if (!(el > 0.0)) {
      for (int c = r; c < m; ++c) rmat[IDX(r, c, m)] = -rmat[IDX(r, c, m)];
      double* qc = qmat + IDX(0, r, n + m);
      for (int64_t i = 0; i < n; ++i) qc[i] = -qc[i];
      el = -el;
    }
    log_det_r += log(el);
  }
  log_det_r += log_det_r;
  const double l1 = -0.5 * (log_det_r - log_det_km + log_det_s + (double)n * LOG_2PI);
  secs[1] = now() - t0;
  t0 = now();

  /* ---- Deriv.Trained.calc (fitc_gp.ml:1158-1181) */
  for (int64_t i = 0; i < n; ++i) y_[i] = y[i] * sqrt_is[i];
  L.gemv("T", &N, &M, &d1, qmat, &NM, y_, &one, &d0, qty, &one);
  memcpy(u_vec, y_, n * 8);
  L.gemv("N", &N, &M, &dm1, qmat, &NM, qty, &one, &d1, u_vec, &one);
  double l2 = 0.0;
  for (int64_t i = 0; i < n; ++i) l2 += u_vec[i] * y_[i];
  l2 *= -0.5;
  memcpy(coeffs, qty, (size_t)m * 8);
  L.trsv("U", "N", "N", &M, rmat, &M, coeffs, &one);
  /* ---- Cm.calc_common / calc_internal (fitc_gp.ml:1037-1078): K_m^-1, B^-1, T, q_diag */
  memcpy(tmat, chol, mm * 8);
  L.potri("U", &M, tmat, &M, &info);
  if (info) return -6;
  memcpy(inv_b, rmat, mm * 8);
  L.potri("U", &M, inv_b, &M, &info);
  if (info) return -7;
  for (int c = 0; c < m; ++c)
    for (int r = 0; r < m; ++r) tmat[IDX(r, c, m)] = (r <= c) ? tmat[IDX(r, c, m)] - inv_b[IDX(r, c, m)] : 0.0;
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < n; ++r) {
    double s = 0.0;
    for (int c = 0; c < m; ++c) s += qmat[IDX(r, c, n + m)] * qmat[IDX(r, c, n + m)];
    q_diag[r] = s;
  }
  double sum_v = 0.0;
  for (int64_t i = 0; i < n; ++i) { /* fitc_gp.ml:1092-1108, :1164-1175 */
    w_vec[i] = u_vec[i] * sqrt_is[i];
    v1_vec[i] = is_vec[i] * (1.0 - q_diag[i]);
    v_vec[i] = v1_vec[i] - w_vec[i] * w_vec[i];
    sq_v1[i] = sqrt(v1_vec[i]);
    sum_v += v_vec[i];
  }
  const double dlds2 = -0.5 * sum_v; /* fitc_gp.ml:1112-1119 */
  secs[2] = now() - t0;
  t0 = now();

  /* ---- Shared.calc_us_mat (fitc_gp.ml:931-939) */
  memcpy(umat, vmat, nm * 8);
  L.trsm("R", "U", "T", "N", &N, &M, &d1, chol, &M, umat, &N);
#pragma omp parallel for schedule(static)
  for (int c = 0; c < m; ++c) memcpy(xmat + IDX(0, c, n), qmat + IDX(0, c, n + m), n * 8);
  L.trsm("R", "U", "T", "N", &N, &M, &d1, rmat, &M, xmat, &N);
  /* ---- Trained.prepare_hyper (fitc_gp.ml:1192-1207): W = T - t t^T - U1^T U1 + U2^T U2; X = S - diag(v) U - w t^T */
  for (int c = 0; c < m; ++c)
    for (int r = 0; r < m; ++r) wmat[IDX(r, c, m)] = (r <= c) ? tmat[IDX(r, c, m)] - coeffs[r] * coeffs[c] : 0.0;
#pragma omp parallel for schedule(static)
  for (int c = 0; c < m; ++c)
    for (int64_t r = 0; r < n; ++r) u1[IDX(r, c, n)] = umat[IDX(r, c, n)] * sq_v1[r];
  L.syrk("U", "T", &M, &N, &dm1, u1, &N, &d1, wmat, &M);
#pragma omp parallel for schedule(static)
  for (int c = 0; c < m; ++c)
    for (int64_t r = 0; r < n; ++r) u1[IDX(r, c, n)] = umat[IDX(r, c, n)] * w_vec[r];
  L.syrk("U", "T", &M, &N, &d1, u1, &N, &d1, wmat, &M);
#pragma omp parallel for schedule(static)
  for (int c = 0; c < m; ++c) {
    double* xc = xmat + IDX(0, c, n);
    const double* uc = umat + IDX(0, c, n);
    const double tc = coeffs[c];
    for (int64_t r = 0; r < n; ++r) xc[r] = xc[r] * sqrt_is[r] - v_vec[r] * uc[r] - w_vec[r] * tc;
  }
  secs[3] = now() - t0;
  t0 = now();

  /* ---- per-hyper log-evidence derivatives (fitc_gp.ml:1005-1021 over cov_se_iso.ml:247-327) */
  /* Log_ell: dkn_diag `Const 0; dkm `Dense K .* sqdiff * inv_ell2 (diagonal 0), Mat.symm2_trace; dknm `Dense */
  double tr_wkd = 0.0, tr_wk = 0.0, sum_xk = 0.0, sum_xkd = 0.0;
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : tr_wkd, tr_wk)
  for (int c = 0; c < m; ++c) {
    for (int r = 0; r < c; ++r) {
      const double wk = wmat[IDX(r, c, m)] * km[IDX(r, c, m)];
      tr_wk += 2.0 * wk;
      tr_wkd += 2.0 * wk * sd_m[IDX(r, c, m)] * inv_ell2;
    }
    tr_wk += wmat[IDX(c, c, m)] * km[IDX(c, c, m)];
  }
#pragma omp parallel for schedule(static) reduction(+ : sum_xk, sum_xkd)
  for (int c = 0; c < m; ++c) {
    const double* xc = xmat + IDX(0, c, n);
    const double* kc = knm + IDX(0, c, n);
    const double* sc = sd_nm + IDX(0, c, n);
    for (int64_t r = 0; r < n; ++r) {
      const double xk = xc[r] * kc[r];
      sum_xk += xk;
      sum_xkd += xk * sc[r] * inv_ell2;
    }
  }
  grad[0] = -0.5 * (0.0 - tr_wkd) - sum_xkd;
  /* Log_sf2: `Factor 1 on all three */
  grad[1] = -0.5 * (sf2 * sum_v - tr_wk) - sum_xk;
  /* Inducing_hyper {ind; dim}: dkm `Sparse_rows (utils.ml:196-220), dknm `Sparse_cols */
#pragma omp parallel for schedule(dynamic, 4)
  for (int ind = 0; ind < m; ++ind) {
    const double* xc = xmat + IDX(0, ind, n);
    const double* kc = knm + IDX(0, ind, n);
    for (int dim = 0; dim < d; ++dim) {
      const double zc = Z[IDX(dim, ind, d)];
      double full = 0.0;
      for (int i = 0; i < m; ++i) {
        if (i == ind) continue;
        const double kel = (i < ind) ? km[IDX(i, ind, m)] : km[IDX(ind, i, m)];
        const double wel = (i < ind) ? wmat[IDX(i, ind, m)] : wmat[IDX(ind, i, m)];
        full += wel * (inv_ell2 * (Z[IDX(dim, i, d)] - zc) * kel);
      }
      const double dkm_term = full + full;
      double dknm_term = 0.0;
      for (int64_t r = 0; r < n; ++r) dknm_term += xc[r] * (inv_ell2 * (X[IDX(dim, r, d)] - zc) * kc[r]);
      grad[2 + (int64_t)ind * d + dim] = 0.5 * dkm_term - dknm_term;
    }
  }
  secs[4] = now() - t0;
  secs[5] = now() - t_start;
  secs[6] = L.get_threads ? (double)L.get_threads() : (double)threads; /* BLAS threads actually in use */
  secs[7] = (double)omp_get_max_threads();
  out[0] = l1;
  out[1] = l2;
  out[2] = l1 + l2;
  out[3] = dlds2;
  free(km); free(sd_m); free(chol); free(knm); free(sd_nm); free(vmat); free(qmat); free(umat); free(xmat); free(u1);
  free(rmat); free(tmat); free(wmat); free(inv_b); free(vecs); free(tau); free(qty);
  return 0;
}
