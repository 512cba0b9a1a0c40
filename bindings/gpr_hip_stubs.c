/* OCaml stubs over the gprhip C ABI (include/gprhip.h) -- the binding a maintainer of mmottl/gpr adds next to
 * lib/.  NOT COMPILED IN THIS REPOSITORY: the build image has no OCaml toolchain (no ocaml / dune / opam headers),
 * so this file has never met <caml/...>.  It is written against the OCaml 4.14 / 5.x C interface; the Python mirror
 * (gpr_amd/_lib.py) exercises the same entry points with the same argument conventions and is what the tests run.
 *
 * Conventions: Bigarray.Array2 / Array1 of float64 in Fortran layout arrive as `value`; their data pointers are
 * borrowed for the duration of the call only (the library copies before returning), so no global roots are needed.
 * Blocking calls release the runtime lock.  A non-zero status becomes Failure with the library's message, whose
 * prefixes follow the reference ("Model.check_sigma2: ...", "Lacaml.D.potrf: ...").
 */
#define CAML_NAME_SPACE
#include <caml/alloc.h>
#include <caml/bigarray.h>
#include <caml/custom.h>
#include <caml/fail.h>
#include <caml/memory.h>
#include <caml/mlvalues.h>
#include <caml/threads.h>
#include <string.h>

#include "gprhip.h"

#define Problem_val(v) (*((gprhip_problem**)Data_custom_val(v)))

static void problem_finalize(value v) {
  gprhip_problem* p = Problem_val(v);
  if (p) {
    gprhip_problem_destroy(p);
    Problem_val(v) = NULL;
  }
}
static struct custom_operations problem_ops = {"gprhip.problem",           problem_finalize,
                                               custom_compare_default,     custom_hash_default,
                                               custom_serialize_default,   custom_deserialize_default,
                                               custom_compare_ext_default, custom_fixed_length_default};

static void check(int status) {
  if (status != GPRHIP_OK) caml_failwith(gprhip_last_error());
}
static const double* opt_data(value opt) { /* 'a option of a Bigarray -> data pointer or NULL */
  return Is_block(opt) ? (const double*)Caml_ba_data_val(Field(opt, 0)) : NULL;
}

/* external problem_create : int -> int -> int -> int -> int -> int -> int -> int -> problem
 *   device cov_kind precision n big_d d m chunk_rows */
CAMLprim value gprhip_ml_problem_create(value device, value kind, value precision, value n, value big_d, value d,
                                        value m, value chunk_rows) {
  CAMLparam0();
  CAMLlocal1(res);
  gprhip_problem* p = NULL;
  check(gprhip_problem_create_ex(Int_val(device), Int_val(kind), Int_val(precision), (int64_t)Long_val(n),
                                 Int_val(big_d), Int_val(d), Int_val(m), (int64_t)Long_val(chunk_rows), &p));
  res = caml_alloc_custom(&problem_ops, sizeof(gprhip_problem*), 0, 1);
  Problem_val(res) = p;
  CAMLreturn(res);
}
CAMLprim value gprhip_ml_problem_create_bc(value* a, int n) {
  (void)n;
  return gprhip_ml_problem_create(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7]);
}

CAMLprim value gprhip_ml_problem_destroy(value prob) {
  problem_finalize(prob);
  return Val_unit;
}

/* external set_inputs : problem -> mat -> unit      (Fortran D x n: leading dimension = dim1) */
CAMLprim value gprhip_ml_set_inputs(value prob, value mat) {
  CAMLparam2(prob, mat);
  struct caml_ba_array* ba = Caml_ba_array_val(mat);
  check(gprhip_set_inputs(Problem_val(prob), (const double*)ba->data, (int64_t)ba->dim[0]));
  CAMLreturn(Val_unit);
}
CAMLprim value gprhip_ml_set_targets(value prob, value vec) {
  CAMLparam2(prob, vec);
  check(gprhip_set_targets(Problem_val(prob), (const double*)Caml_ba_data_val(vec)));
  CAMLreturn(Val_unit);
}

/* The hyper-parameter record of gpr_hip.ml:
 *   { log_ell; log_sf2; sigma2; inducing : mat; tproj : mat option; variational : bool; model_only : bool;
 *     jitter; log_hetero_skedasticity : vec option; log_multiscales_m05 : mat option; reuse_v : bool }
 * (a record whose float fields are boxed: it has non-float fields). */
static void hypers_of_value(value h, gprhip_hypers* out) {
  memset(out, 0, sizeof *out);
  out->log_ell = Double_val(Field(h, 0));
  out->log_sf2 = Double_val(Field(h, 1));
  out->sigma2 = Double_val(Field(h, 2));
  out->inducing = (const double*)Caml_ba_data_val(Field(h, 3));
  out->tproj = opt_data(Field(h, 4));
  out->variational = Bool_val(Field(h, 5));
  out->model_only = Bool_val(Field(h, 6));
  out->jitter = Double_val(Field(h, 7));
  out->log_hetero_skedasticity = opt_data(Field(h, 8));
  out->log_multiscales_m05 = opt_data(Field(h, 9));
  out->reuse_v = Bool_val(Field(h, 10));
}

/* external eval : problem -> hypers -> want_grad:bool -> grad:vec -> coeffs:vec -> float * float * float * float * int
 *   returns (l1, l2, l, dl_dsigma2, n_hypers); fills grad (Hyper.get_all order) and coeffs (m) */
CAMLprim value gprhip_ml_eval(value prob, value h, value want_grad, value grad, value coeffs) {
  CAMLparam5(prob, h, want_grad, grad, coeffs);
  CAMLlocal1(res);
  gprhip_hypers hy;
  gprhip_result r;
  gprhip_problem* p = Problem_val(prob);
  double* g = (double*)Caml_ba_data_val(grad);
  double* c = (double*)Caml_ba_data_val(coeffs);
  int wg = Bool_val(want_grad), status;
  hypers_of_value(h, &hy);
  /* the Bigarray payloads live outside the OCaml heap: safe to touch without the runtime lock */
  caml_release_runtime_system();
  status = gprhip_eval(p, &hy, wg, &r, g, c);
  caml_acquire_runtime_system();
  check(status);
  res = caml_alloc_tuple(5);
  Store_field(res, 0, caml_copy_double(r.l1));
  Store_field(res, 1, caml_copy_double(r.l2));
  Store_field(res, 2, caml_copy_double(r.l));
  Store_field(res, 3, caml_copy_double(r.dl_dsigma2));
  Store_field(res, 4, Val_long(r.n_hypers));
  CAMLreturn(res);
}

/* external n_hypers : problem -> int -> int     flags: 1 tproj, 2 hetero, 4 multiscales */
CAMLprim value gprhip_ml_n_hypers(value prob, value flags) {
  return Val_long(gprhip_n_hypers(Problem_val(prob), Int_val(flags)));
}

/* external predict : problem -> mat -> predictive:bool -> means:vec -> variances:vec option -> unit */
CAMLprim value gprhip_ml_predict(value prob, value points, value predictive, value means, value variances) {
  CAMLparam5(prob, points, predictive, means, variances);
  struct caml_ba_array* ba = Caml_ba_array_val(points);
  gprhip_problem* p = Problem_val(prob);
  const double* x = (const double*)ba->data;
  int64_t ld = ba->dim[0], nt = ba->dim[1];
  double* mu = (double*)Caml_ba_data_val(means);
  double* var = (double*)opt_data(variances);
  int pr = Bool_val(predictive), status;
  caml_release_runtime_system();
  status = gprhip_predict(p, x, ld, nt, pr, mu, var);
  caml_acquire_runtime_system();
  check(status);
  CAMLreturn(Val_unit);
}

/* external train_stats : problem -> means:vec option -> sums:vec -> unit      sums = [sse; sum|e|; max|e|; sum y^2] */
CAMLprim value gprhip_ml_train_stats(value prob, value means, value sums) {
  CAMLparam3(prob, means, sums);
  check(gprhip_train_stats(Problem_val(prob), (double*)opt_data(means), (double*)Caml_ba_data_val(sums)));
  CAMLreturn(Val_unit);
}

/* external covariances : problem -> mat -> kind:int -> predictive:bool -> cov:mat -> unit   kind 0 FITC, 1 FIC */
CAMLprim value gprhip_ml_covariances(value prob, value points, value kind, value predictive, value cov) {
  CAMLparam5(prob, points, kind, predictive, cov);
  struct caml_ba_array* ba = Caml_ba_array_val(points);
  check(gprhip_covariances(Problem_val(prob), (const double*)ba->data, (int64_t)ba->dim[0], (int64_t)ba->dim[1],
                           Int_val(kind), Bool_val(predictive), (double*)Caml_ba_data_val(cov)));
  CAMLreturn(Val_unit);
}

/* external cov_samples : problem -> cov:mat -> add_diag:float -> jitter:float -> means:vec -> z:mat -> samples:mat -> unit */
CAMLprim value gprhip_ml_cov_samples(value prob, value cov, value add_diag, value jitter, value means, value z,
                                     value samples) {
  CAMLparam5(prob, cov, add_diag, jitter, means);
  CAMLxparam2(z, samples);
  struct caml_ba_array* c = Caml_ba_array_val(cov);
  struct caml_ba_array* zz = Caml_ba_array_val(z);
  check(gprhip_cov_samples(Problem_val(prob), (const double*)c->data, (int64_t)c->dim[0], (int64_t)c->dim[1],
                           Double_val(add_diag), Double_val(jitter), (const double*)Caml_ba_data_val(means),
                           (const double*)zz->data, (int64_t)zz->dim[1], (double*)Caml_ba_data_val(samples)));
  CAMLreturn(Val_unit);
}
CAMLprim value gprhip_ml_cov_samples_bc(value* a, int n) {
  (void)n;
  return gprhip_ml_cov_samples(a[0], a[1], a[2], a[3], a[4], a[5], a[6]);
}

/* external co_variance_coeffs : problem -> chol_km:mat -> r_mat:mat -> unit */
CAMLprim value gprhip_ml_co_variance_coeffs(value prob, value chol_km, value r_mat) {
  CAMLparam3(prob, chol_km, r_mat);
  check(gprhip_co_variance_coeffs(Problem_val(prob), (double*)Caml_ba_data_val(chol_km),
                                  (double*)Caml_ba_data_val(r_mat)));
  CAMLreturn(Val_unit);
}

/* external load_predictor : problem -> hypers -> coeffs:vec option -> (mat * mat) option -> unit */
CAMLprim value gprhip_ml_load_predictor(value prob, value h, value coeffs, value factors) {
  CAMLparam4(prob, h, coeffs, factors);
  gprhip_hypers hy;
  const double *u = NULL, *r = NULL;
  hypers_of_value(h, &hy);
  if (Is_block(factors)) {
    u = (const double*)Caml_ba_data_val(Field(Field(factors, 0), 0));
    r = (const double*)Caml_ba_data_val(Field(Field(factors, 0), 1));
  }
  check(gprhip_load_predictor(Problem_val(prob), &hy, opt_data(coeffs), u, r));
  CAMLreturn(Val_unit);
}

/* row-sharded evaluation (one process per GPU; the exchange buffers are device memory owned by the caller's
 * collective library, passed as nativeint addresses) */
CAMLprim value gprhip_ml_ar_len(value prob, value which) {
  return Val_long(Int_val(which) == 1 ? gprhip_ar1_len(Problem_val(prob)) : gprhip_ar2_len(Problem_val(prob)));
}
CAMLprim value gprhip_ml_eval_pass1(value prob, value h, value want_grad, value n_total, value d_ar1) {
  CAMLparam5(prob, h, want_grad, n_total, d_ar1);
  gprhip_hypers hy;
  hypers_of_value(h, &hy);
  check(gprhip_eval_pass1(Problem_val(prob), &hy, Bool_val(want_grad), (int64_t)Long_val(n_total),
                          (double*)Nativeint_val(d_ar1)));
  CAMLreturn(Val_unit);
}
CAMLprim value gprhip_ml_eval_pass2(value prob, value d_ar1, value d_ar2) {
  check(gprhip_eval_pass2(Problem_val(prob), (const double*)Nativeint_val(d_ar1), (double*)Nativeint_val(d_ar2)));
  return Val_unit;
}
CAMLprim value gprhip_ml_eval_finish(value prob, value d_ar2, value grad, value coeffs) {
  CAMLparam4(prob, d_ar2, grad, coeffs);
  CAMLlocal1(res);
  gprhip_result r;
  check(gprhip_eval_finish(Problem_val(prob), (const double*)Nativeint_val(d_ar2), &r,
                           (double*)Caml_ba_data_val(grad), (double*)Caml_ba_data_val(coeffs)));
  res = caml_alloc_tuple(5);
  Store_field(res, 0, caml_copy_double(r.l1));
  Store_field(res, 1, caml_copy_double(r.l2));
  Store_field(res, 2, caml_copy_double(r.l));
  Store_field(res, 3, caml_copy_double(r.dl_dsigma2));
  Store_field(res, 4, Val_long(r.n_hypers));
  CAMLreturn(res);
}
CAMLprim value gprhip_ml_sync(value prob) {
  check(gprhip_sync(Problem_val(prob)));
  return Val_unit;
}
CAMLprim value gprhip_ml_stream(value prob) { return caml_copy_nativeint((intnat)gprhip_stream(Problem_val(prob))); }
CAMLprim value gprhip_ml_set_timing(value prob, value level) {
  check(gprhip_set_timing(Problem_val(prob), Int_val(level)));
  return Val_unit;
}
CAMLprim value gprhip_ml_device_count(value unit) {
  int c = 0;
  (void)unit;
  check(gprhip_device_count(&c));
  return Val_int(c);
}
CAMLprim value gprhip_ml_version(value unit) {
  (void)unit;
  return caml_copy_string(gprhip_version());
}

/* ---- single-process, multi-device: gprhip_ctx_* / gprhip_sharded_* (include/gprhip.h) ------------------------------
 * The reference's host is one process (bin/ocaml_gpr.ml:176-177, :340-342), so this is the path its functor takes:
 * a context over the node's devices, one sharded problem per set of training inputs, RCCL inside the library. */
#define Ctx_val(v) (*((gprhip_ctx**)Data_custom_val(v)))
#define Sharded_val(v) (*((gprhip_sharded**)Data_custom_val(v)))

static void ctx_finalize(value v) {
  gprhip_ctx* c = Ctx_val(v);
  if (c) {
    gprhip_ctx_destroy(c); /* deferred by the library while sharded problems of it are alive */
    Ctx_val(v) = NULL;
  }
}
static void sharded_finalize(value v) {
  gprhip_sharded* s = Sharded_val(v);
  if (s) {
    gprhip_sharded_destroy(s);
    Sharded_val(v) = NULL;
  }
}
static struct custom_operations ctx_ops = {"gprhip.ctx",
                                           ctx_finalize,
                                           custom_compare_default,
                                           custom_hash_default,
                                           custom_serialize_default,
                                           custom_deserialize_default,
                                           custom_compare_ext_default,
                                           custom_fixed_length_default};
static struct custom_operations sharded_ops = {"gprhip.sharded",
                                               sharded_finalize,
                                               custom_compare_default,
                                               custom_hash_default,
                                               custom_serialize_default,
                                               custom_deserialize_default,
                                               custom_compare_ext_default,
                                               custom_fixed_length_default};
/* a shard's device problem, owned by its sharded problem: no finaliser (the OCaml side keeps the owner reachable) */
static struct custom_operations borrowed_problem_ops = {"gprhip.problem.borrowed",
                                                        custom_finalize_default,
                                                        custom_compare_default,
                                                        custom_hash_default,
                                                        custom_serialize_default,
                                                        custom_deserialize_default,
                                                        custom_compare_ext_default,
                                                        custom_fixed_length_default};

/* external ctx_create : int array -> ctx */
CAMLprim value gprhip_ml_ctx_create(value devices) {
  CAMLparam1(devices);
  CAMLlocal1(res);
  int devs[64], n = (int)Wosize_val(devices), i;
  gprhip_ctx* c = NULL;
  if (n < 1 || n > 64) caml_invalid_argument("Gpr_hip.ctx_create: 1 to 64 devices");
  for (i = 0; i < n; ++i) devs[i] = Int_val(Field(devices, i));
  check(gprhip_ctx_create(devs, n, &c));
  res = caml_alloc_custom(&ctx_ops, sizeof(gprhip_ctx*), 0, 1);
  Ctx_val(res) = c;
  CAMLreturn(res);
}
CAMLprim value gprhip_ml_ctx_destroy(value ctx) {
  ctx_finalize(ctx);
  return Val_unit;
}
CAMLprim value gprhip_ml_ctx_ndev(value ctx) { return Val_int(gprhip_ctx_ndev(Ctx_val(ctx))); }
CAMLprim value gprhip_ml_ctx_comm_mode(value ctx) { return Val_int(gprhip_ctx_comm_mode(Ctx_val(ctx))); }

/* external sharded_create : ctx -> int -> int -> int -> int -> int -> int -> int -> sharded
 *   cov_kind precision n big_d d m chunk_rows */
CAMLprim value gprhip_ml_sharded_create(value ctx, value kind, value precision, value n, value big_d, value d,
                                        value m, value chunk_rows) {
  CAMLparam1(ctx);
  CAMLlocal1(res);
  gprhip_sharded* s = NULL;
  check(gprhip_sharded_create(Ctx_val(ctx), Int_val(kind), Int_val(precision), (int64_t)Long_val(n), Int_val(big_d),
                              Int_val(d), Int_val(m), (int64_t)Long_val(chunk_rows), &s));
  res = caml_alloc_custom(&sharded_ops, sizeof(gprhip_sharded*), 0, 1);
  Sharded_val(res) = s;
  CAMLreturn(res);
}
CAMLprim value gprhip_ml_sharded_create_bc(value* a, int n) {
  (void)n;
  return gprhip_ml_sharded_create(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7]);
}
CAMLprim value gprhip_ml_sharded_destroy(value sp) {
  sharded_finalize(sp);
  return Val_unit;
}

/* external sharded_shard : sharded -> int -> int * int * int       (device, row_lo, row_hi) of a shard */
CAMLprim value gprhip_ml_sharded_shard(value sp, value idx) {
  CAMLparam1(sp);
  CAMLlocal1(res);
  int dev = 0;
  int64_t lo = 0, hi = 0;
  check(gprhip_sharded_shard(Sharded_val(sp), Int_val(idx), &dev, &lo, &hi));
  res = caml_alloc_tuple(3);
  Store_field(res, 0, Val_int(dev));
  Store_field(res, 1, Val_long(lo));
  Store_field(res, 2, Val_long(hi));
  CAMLreturn(res);
}
/* external shard_rows : n:int -> ndev:int -> idx:int -> int * int      (the partition itself; no device involved) */
CAMLprim value gprhip_ml_shard_rows(value n, value ndev, value idx) {
  CAMLparam0();
  CAMLlocal1(res);
  int64_t lo = 0, hi = 0;
  check(gprhip_shard_rows((int64_t)Long_val(n), Int_val(ndev), Int_val(idx), &lo, &hi));
  res = caml_alloc_tuple(2);
  Store_field(res, 0, Val_long(lo));
  Store_field(res, 1, Val_long(hi));
  CAMLreturn(res);
}

/* external sharded_problem : sharded -> int -> problem     (borrowed: valid while the sharded problem lives) */
CAMLprim value gprhip_ml_sharded_problem(value sp, value idx) {
  CAMLparam1(sp);
  CAMLlocal1(res);
  gprhip_problem* p = gprhip_sharded_problem(Sharded_val(sp), Int_val(idx));
  if (!p) caml_invalid_argument("Gpr_hip.sharded_problem: no such shard");
  res = caml_alloc_custom(&borrowed_problem_ops, sizeof(gprhip_problem*), 0, 1);
  Problem_val(res) = p;
  CAMLreturn(res);
}

CAMLprim value gprhip_ml_sharded_set_inputs(value sp, value mat) {
  CAMLparam2(sp, mat);
  struct caml_ba_array* ba = Caml_ba_array_val(mat);
  check(gprhip_sharded_set_inputs(Sharded_val(sp), (const double*)ba->data, (int64_t)ba->dim[0]));
  CAMLreturn(Val_unit);
}
CAMLprim value gprhip_ml_sharded_set_targets(value sp, value vec) {
  CAMLparam2(sp, vec);
  check(gprhip_sharded_set_targets(Sharded_val(sp), (const double*)Caml_ba_data_val(vec)));
  CAMLreturn(Val_unit);
}

/* external sharded_eval : sharded -> hypers -> want_grad:bool -> grad:vec -> coeffs:vec
 *                         -> float * float * float * float * int        as gprhip_ml_eval */
CAMLprim value gprhip_ml_sharded_eval(value sp, value h, value want_grad, value grad, value coeffs) {
  CAMLparam5(sp, h, want_grad, grad, coeffs);
  CAMLlocal1(res);
  gprhip_hypers hy;
  gprhip_result r;
  gprhip_sharded* s = Sharded_val(sp);
  double* g = (double*)Caml_ba_data_val(grad);
  double* c = (double*)Caml_ba_data_val(coeffs);
  int wg = Bool_val(want_grad), status;
  hypers_of_value(h, &hy);
  caml_release_runtime_system();
  status = gprhip_sharded_eval(s, &hy, wg, &r, g, c);
  caml_acquire_runtime_system();
  check(status);
  res = caml_alloc_tuple(5);
  Store_field(res, 0, caml_copy_double(r.l1));
  Store_field(res, 1, caml_copy_double(r.l2));
  Store_field(res, 2, caml_copy_double(r.l));
  Store_field(res, 3, caml_copy_double(r.dl_dsigma2));
  Store_field(res, 4, Val_long(r.n_hypers));
  CAMLreturn(res);
}

/* external sharded_predict : sharded -> mat -> predictive:bool -> means:vec -> variances:vec option -> unit */
CAMLprim value gprhip_ml_sharded_predict(value sp, value points, value predictive, value means, value variances) {
  CAMLparam5(sp, points, predictive, means, variances);
  struct caml_ba_array* ba = Caml_ba_array_val(points);
  gprhip_sharded* s = Sharded_val(sp);
  const double* x = (const double*)ba->data;
  int64_t ld = ba->dim[0], nt = ba->dim[1];
  double* mu = (double*)Caml_ba_data_val(means);
  double* var = (double*)opt_data(variances);
  int pr = Bool_val(predictive), status;
  caml_release_runtime_system();
  status = gprhip_sharded_predict(s, x, ld, nt, pr, mu, var);
  caml_acquire_runtime_system();
  check(status);
  CAMLreturn(Val_unit);
}
/* external sharded_train_stats : sharded -> means:vec option -> sums:vec -> unit */
CAMLprim value gprhip_ml_sharded_train_stats(value sp, value means, value sums) {
  CAMLparam3(sp, means, sums);
  check(gprhip_sharded_train_stats(Sharded_val(sp), (double*)opt_data(means), (double*)Caml_ba_data_val(sums)));
  CAMLreturn(Val_unit);
}

/* external sharded_comm_stats : sharded -> int * (int * int) * (float * float)   collectives, bytes, milliseconds */
CAMLprim value gprhip_ml_sharded_comm_stats(value sp) {
  CAMLparam1(sp);
  CAMLlocal3(res, b, t);
  int k = 0;
  int64_t bytes[2] = {0, 0};
  float ms[2] = {0.f, 0.f};
  check(gprhip_sharded_comm_stats(Sharded_val(sp), &k, bytes, ms));
  b = caml_alloc_tuple(2);
  Store_field(b, 0, Val_long(bytes[0]));
  Store_field(b, 1, Val_long(bytes[1]));
  t = caml_alloc_tuple(2);
  Store_field(t, 0, caml_copy_double(ms[0]));
  Store_field(t, 1, caml_copy_double(ms[1]));
  res = caml_alloc_tuple(3);
  Store_field(res, 0, Val_int(k));
  Store_field(res, 1, b);
  Store_field(res, 2, t);
  CAMLreturn(res);
}
CAMLprim value gprhip_ml_sharded_set_timing(value sp, value level) {
  check(gprhip_sharded_set_timing(Sharded_val(sp), Int_val(level)));
  return Val_unit;
}

/* external condition : problem -> float * float       (cond estimate of K_m + jitter, mean-coefficient error bound) */
CAMLprim value gprhip_ml_condition(value prob) {
  CAMLparam1(prob);
  CAMLlocal1(res);
  double c = 0.0, b = 0.0;
  check(gprhip_condition(Problem_val(prob), &c, &b));
  res = caml_alloc_tuple(2);
  Store_field(res, 0, caml_copy_double(c));
  Store_field(res, 1, caml_copy_double(b));
  CAMLreturn(res);
}

/* external debug_fetch : problem -> string -> vec -> unit      intermediates of the last evaluation (parity tests) */
CAMLprim value gprhip_ml_debug_fetch(value prob, value name, value out) {
  CAMLparam3(prob, name, out);
  check(gprhip_debug_fetch(Problem_val(prob), String_val(name), (double*)Caml_ba_data_val(out),
                           (int64_t)Caml_ba_array_val(out)->dim[0]));
  CAMLreturn(Val_unit);
}
