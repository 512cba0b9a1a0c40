// gprhip.hpp -- C++ host mirror of the reference's functor surface over the C ABI of gprhip.h.
//
// The reference is compiled OCaml:  module GP = Fitc_gp.Make_deriv (Cov_se_fat.Deriv)  gives  GP.FITC,
// GP.Variational_FITC, GP.FIC, GP.Variational_FIC, each with an Eval and a Deriv side (lib/fitc_gp.mli:83-135,
// lib/interfaces.ml Sigs.Eval :373-844, Sigs.Deriv :848-1154).  This header is that surface for a C++ caller:
//
//     using GP = gpr::Make_deriv<gpr::Cov_se_iso>;                 // Fitc_gp.Make_deriv (Cov_se_iso.Deriv)
//     auto kernel   = gpr::Cov_se_iso::Kernel::create({log_ell, log_sf2});
//     auto inducing = GP::FITC::Inducing::calc(kernel, inducing_points);
//     auto inputs   = GP::FITC::Inputs::calc(inducing, training_inputs);
//     auto model    = GP::FITC::Model::calc(inputs, sigma2);
//     auto trained  = GP::FITC::Trained::calc(model, targets);
//     double l      = GP::FITC::Trained::calc_log_evidence(trained);
//     auto hyper_t  = GP::FITC::Trained::prepare_hyper(trained);
//     double dl     = GP::FITC::Trained::calc_log_evidence(hyper_t, hyper);
//
// Inputs.t, Model.t, Trained.t and hyper_t are abstract in the reference signature (lib/interfaces.ml:433, :459,
// :514, :895-899, :944-948); here they are small value types sharing one device-resident problem.  The reference
// computes stage by stage; the device path is a two-pass streaming evaluation, so these objects are lazy and the
// evaluation runs when the first number is asked for.  Errors: the reference raises Failure / Invalid_argument;
// this header throws gpr::Failure carrying the library's message and status.
//
// Header-only, C++17, links against libgprhip.so.  Matrices are column-major like the reference's Bigarrays.
#ifndef GPRHIP_HPP
#define GPRHIP_HPP

#include <cmath>
#include <cstdint>
#include <map>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "gprhip.h"

namespace gpr {

struct Failure : std::runtime_error {
  int status;
  Failure(int st, const std::string& msg) : std::runtime_error(msg), status(st) {}
};
inline void check(int status) {
  if (status != GPRHIP_OK) throw Failure(status, gprhip_last_error());
}

using Vec = std::vector<double>;
// Column-major matrix (Lacaml.D.mat): element (r, c), 0-based, at a[c * rows + r].
struct Mat {
  int rows = 0, cols = 0;
  std::vector<double> a;
  Mat() = default;
  Mat(int r, int c, double v = 0.0) : rows(r), cols(c), a((size_t)r * c, v) {}
  double& operator()(int r, int c) { return a[(size_t)c * rows + r]; }
  double operator()(int r, int c) const { return a[(size_t)c * rows + r]; }
  const double* data() const { return a.data(); }
  double* data() { return a.data(); }
};
using MatP = std::shared_ptr<const Mat>;  // physical identity stands in for the reference's phys_equal checks

constexpr double cholesky_jitter = 1e-6;  // Utils.cholesky_jitter, lib/utils.ml:35

// One hyper-parameter of either covariance (Cov_se_iso.Hyper.t lib/cov_se_iso.ml:27-28, Cov_se_fat.Hyper.t
// lib/cov_se_fat.ml:258-284); ind / dim / big_dim / small_dim are 1-based like the reference.
struct Hyper {
  enum Kind { Log_ell, Log_sf2, Inducing_hyper, Proj, Log_hetero_skedasticity, Log_multiscale_m05 } kind;
  int ind = 0, dim = 0;  // Inducing_hyper{ind; dim}, Log_multiscale_m05{ind; dim}, Log_hetero_skedasticity dim
  int big_dim = 0, small_dim = 0;  // Proj{big_dim; small_dim}
};

// ---- Cov_se_iso (lib/cov_se_iso.ml) -----------------------------------------------------------------------
struct Cov_se_iso {
  static constexpr int cov_kind = GPRHIP_COV_SE_ISO;
  struct Params { double log_ell = 0.0, log_sf2 = 0.0; };  // :23-25; create_default_kernel_params :122-123
  struct Kernel {
    Params params;
    double inv_ell2 = 1.0, inv_ell2_05 = -0.5, log_sf2 = 0.0, sf2 = 1.0;
    static Kernel create(const Params& p) {  // :41-44
      Kernel k;
      k.params = p;
      k.inv_ell2 = std::exp(-2.0 * p.log_ell);
      k.inv_ell2_05 = -0.5 * k.inv_ell2;
      k.log_sf2 = p.log_sf2;
      k.sf2 = std::exp(p.log_sf2);
      return k;
    }
  };
  static int kernel_space_dim(const Kernel&, const Mat& inputs) { return inputs.rows; }
  static void fill(const Kernel& k, gprhip_hypers& h) {
    h.log_ell = k.params.log_ell;
    h.log_sf2 = k.params.log_sf2;
  }
  static int flags(const Kernel&) { return 0; }
  static std::vector<Hyper> get_all(const Kernel&, const Mat& inducing) {  // :188-202
    std::vector<Hyper> hs{{Hyper::Log_ell}, {Hyper::Log_sf2}};
    for (int ind = 1; ind <= inducing.cols; ++ind)
      for (int dim = 1; dim <= inducing.rows; ++dim) hs.push_back({Hyper::Inducing_hyper, ind, dim});
    return hs;
  }
  static double get_value(const Kernel& k, const Mat& inducing, const Hyper& h) {  // :204-207
    if (h.kind == Hyper::Log_ell) return k.params.log_ell;
    if (h.kind == Hyper::Log_sf2) return k.params.log_sf2;
    if (h.kind != Hyper::Inducing_hyper) throw Failure(GPRHIP_EBADARG, "Cov_se_iso.Hyper.get_value: foreign hyper");
    return inducing(h.dim - 1, h.ind - 1);
  }
  // :209-229 (the inducing matrix is copied only if one of its entries is set)
  static std::pair<Kernel, MatP> set_values(const Kernel& k, const MatP& inducing, const std::vector<Hyper>& hs,
                                            const Vec& values) {
    Params p = k.params;
    std::shared_ptr<Mat> z;
    for (size_t i = 0; i < hs.size(); ++i) {
      if (hs[i].kind == Hyper::Log_ell) p.log_ell = values[i];
      else if (hs[i].kind == Hyper::Log_sf2) p.log_sf2 = values[i];
      else {
        if (!z) z = std::make_shared<Mat>(*inducing);
        (*z)(hs[i].dim - 1, hs[i].ind - 1) = values[i];
      }
    }
    return {Kernel::create(p), z ? MatP(z) : inducing};
  }
  static int64_t index_of(const Kernel&, const Mat& inducing, const Hyper& h) {  // position in get_all
    if (h.kind == Hyper::Log_ell) return 0;
    if (h.kind == Hyper::Log_sf2) return 1;
    return 2 + (int64_t)(h.ind - 1) * inducing.rows + (h.dim - 1);
  }
};

// ---- Cov_se_fat (lib/cov_se_fat.ml) -----------------------------------------------------------------------
struct Cov_se_fat {
  static constexpr int cov_kind = GPRHIP_COV_SE_FAT;
  struct Params {  // :27-49
    int d = 0;
    double log_sf2 = 0.0;
    std::optional<Mat> tproj;                    // big_dim x d
    std::optional<Vec> log_hetero_skedasticity;  // m
    std::optional<Mat> log_multiscales_m05;      // d x m
    static Params create(Params p) {             // :38-48
      if (p.tproj && p.tproj->cols != p.d)
        throw Failure(GPRHIP_EBADARG, "Cov_se_fat.Params.create: tproj projection (" + std::to_string(p.tproj->cols) +
                                          ") disagrees with target dimension d (" + std::to_string(p.d) + ")");
      return p;
    }
  };
  struct Kernel {
    Params params;
    double sf2 = 1.0;
    static Kernel create(const Params& p) { return Kernel{p, std::exp(p.log_sf2)}; }  // :62-75
  };
  static int kernel_space_dim(const Kernel& k, const Mat&) { return k.params.d; }
  static void fill(const Kernel& k, gprhip_hypers& h) {
    h.log_sf2 = k.params.log_sf2;
    h.tproj = k.params.tproj ? k.params.tproj->data() : nullptr;
    h.log_hetero_skedasticity = k.params.log_hetero_skedasticity ? k.params.log_hetero_skedasticity->data() : nullptr;
    h.log_multiscales_m05 = k.params.log_multiscales_m05 ? k.params.log_multiscales_m05->data() : nullptr;
  }
  static int flags(const Kernel& k) {
    return (k.params.tproj ? 1 : 0) | (k.params.log_hetero_skedasticity ? 2 : 0) |
           (k.params.log_multiscales_m05 ? 4 : 0);
  }
  static std::vector<Hyper> get_all(const Kernel& k, const Mat& inducing) {  // :290-342
    const int d = k.params.d, m = inducing.cols;
    std::vector<Hyper> hs{{Hyper::Log_sf2}};
    for (int ind = 1; ind <= m; ++ind)
      for (int dim = 1; dim <= d; ++dim) hs.push_back({Hyper::Inducing_hyper, ind, dim});
    if (k.params.tproj)
      for (int big = 1; big <= k.params.tproj->rows; ++big)
        for (int small = 1; small <= d; ++small) hs.push_back({Hyper::Proj, 0, 0, big, small});
    if (k.params.log_hetero_skedasticity)
      for (int i = 1; i <= m; ++i) hs.push_back({Hyper::Log_hetero_skedasticity, 0, i});
    if (k.params.log_multiscales_m05)
      for (int ind = 1; ind <= m; ++ind)
        for (int dim = 1; dim <= d; ++dim) hs.push_back({Hyper::Log_multiscale_m05, ind, dim});
    return hs;
  }
  static double get_value(const Kernel& k, const Mat& inducing, const Hyper& h) {  // :344-362
    auto missing = [](const char* what) {
      return Failure(GPRHIP_EBADARG, std::string("Deriv.Hyper.option_get_value: ") + what + " not supported");
    };
    switch (h.kind) {
      case Hyper::Log_sf2: return k.params.log_sf2;
      case Hyper::Inducing_hyper: return inducing(h.dim - 1, h.ind - 1);
      case Hyper::Proj:
        if (!k.params.tproj) throw missing("tproj");
        return (*k.params.tproj)(h.big_dim - 1, h.small_dim - 1);
      case Hyper::Log_hetero_skedasticity:
        if (!k.params.log_hetero_skedasticity) throw missing("log_hetero_skedasticity");
        return (*k.params.log_hetero_skedasticity)[h.dim - 1];
      case Hyper::Log_multiscale_m05:
        if (!k.params.log_multiscales_m05) throw missing("log_multiscales_m05");
        return (*k.params.log_multiscales_m05)(h.dim - 1, h.ind - 1);
      default: throw Failure(GPRHIP_EBADARG, "Cov_se_fat.Hyper.get_value: foreign hyper");
    }
  }
  static std::pair<Kernel, MatP> set_values(const Kernel& k, const MatP& inducing, const std::vector<Hyper>& hs,
                                            const Vec& values) {  // :364-407
    Params p = k.params;
    std::shared_ptr<Mat> z;
    for (size_t i = 0; i < hs.size(); ++i) {
      const Hyper& h = hs[i];
      switch (h.kind) {
        case Hyper::Log_sf2: p.log_sf2 = values[i]; break;
        case Hyper::Inducing_hyper:
          if (!z) z = std::make_shared<Mat>(*inducing);
          (*z)(h.dim - 1, h.ind - 1) = values[i];
          break;
        case Hyper::Proj:
          if (!p.tproj) throw Failure(GPRHIP_EBADARG, "Deriv.Hyper.option_get_value: tproj not supported");
          (*p.tproj)(h.big_dim - 1, h.small_dim - 1) = values[i];
          break;
        case Hyper::Log_hetero_skedasticity:
          if (!p.log_hetero_skedasticity)
            throw Failure(GPRHIP_EBADARG, "Deriv.Hyper.option_get_value: log_hetero_skedasticity not supported");
          (*p.log_hetero_skedasticity)[h.dim - 1] = values[i];
          break;
        case Hyper::Log_multiscale_m05:
          if (!p.log_multiscales_m05)
            throw Failure(GPRHIP_EBADARG, "Deriv.Hyper.option_get_value: log_multiscales_m05 not supported");
          (*p.log_multiscales_m05)(h.dim - 1, h.ind - 1) = values[i];
          break;
        default: throw Failure(GPRHIP_EBADARG, "Cov_se_fat.Hyper.set_values: foreign hyper");
      }
    }
    return {Kernel::create(p), z ? MatP(z) : inducing};
  }
  static int64_t index_of(const Kernel& k, const Mat& inducing, const Hyper& h) {
    const int64_t d = k.params.d, m = inducing.cols;
    const int64_t nproj = k.params.tproj ? (int64_t)k.params.tproj->rows * d : 0;
    const int64_t nhet = k.params.log_hetero_skedasticity ? m : 0;
    switch (h.kind) {
      case Hyper::Log_sf2: return 0;
      case Hyper::Inducing_hyper: return 1 + (h.ind - 1) * d + (h.dim - 1);
      case Hyper::Proj: return 1 + d * m + (h.big_dim - 1) * d + (h.small_dim - 1);
      case Hyper::Log_hetero_skedasticity: return 1 + d * m + nproj + (h.dim - 1);
      case Hyper::Log_multiscale_m05: return 1 + d * m + nproj + nhet + (h.ind - 1) * d + (h.dim - 1);
      default: throw Failure(GPRHIP_EBADARG, "Cov_se_fat.Hyper.index_of: foreign hyper");
    }
  }
};

// Device-resident problem (RAII over gprhip_problem): one per (training inputs, m, d).
class Problem {
 public:
  Problem(int cov_kind, int64_t n, int D, int d, int m, int device = 0, int precision = GPRHIP_F64,
          int64_t chunk_rows = 0)
      : n(n), D(D), d(d), m(m) {
    check(gprhip_problem_create_ex(device, cov_kind, precision, n, D, d, m, chunk_rows, &p_));
  }
  ~Problem() { gprhip_problem_destroy(p_); }
  Problem(const Problem&) = delete;
  Problem& operator=(const Problem&) = delete;
  gprhip_problem* get() const { return p_; }
  const int64_t n;
  const int D, d, m;
  // identities are kept alive while they are remembered, so a recycled address can never match by accident
  std::shared_ptr<const void> state_owner;    // which evaluation's factors the device holds right now
  std::shared_ptr<const void> last_kernel;    // for Model.update_sigma2's re-use of K_nm, V, r
  std::shared_ptr<const void> last_inducing;

 private:
  gprhip_problem* p_ = nullptr;
};

struct Evaluation {  // everything one device evaluation returns
  double l1 = 0, l2 = 0, l = 0, dl_dsigma2 = 0;
  Vec grad;    // Hyper.get_all order
  Vec coeffs;  // Trained.calc_mean_coeffs
  bool has_grad = false;
};

struct Stats_t {  // Stats.t, lib/fitc_gp.ml:304-315
  int64_t n_samples;
  double target_variance, sse, mse, rmse, smse, msll, mad, maxad;
};

// Fitc_gp.Make_deriv (lib/fitc_gp.mli:120-135)
template <class Spec>
struct Make_deriv {
  using Kernel = typename Spec::Kernel;

  struct Inducing_t {  // Eval.Inducing.t
    Kernel kernel;
    MatP points;
  };
  struct Inputs_t {  // Eval.Inputs.t
    Inducing_t inducing;
    MatP points;
    std::shared_ptr<Problem> problem;  // null for inputs that are only predicted at
  };
  struct Model_t {  // Eval.Model.t / Deriv.Model.t
    Inputs_t inputs;
    double sigma2 = 0;
    bool variational = false;
    std::shared_ptr<Kernel> kernel_ref;  // identity of the kernel (shared by update_sigma2 copies)
    std::shared_ptr<std::map<bool, Evaluation>> ev = std::make_shared<std::map<bool, Evaluation>>();
    std::shared_ptr<int> id = std::make_shared<int>(0);  // identity of this model for the device-state bookkeeping
  };
  struct Trained_t {  // Eval.Trained.t / Deriv.Trained.t
    Model_t model;
    Vec targets;
    bool want_grad = false;
    std::shared_ptr<std::optional<Evaluation>> ev = std::make_shared<std::optional<Evaluation>>();
    std::shared_ptr<int> id = std::make_shared<int>(0);
  };
  struct Hyper_t {  // Deriv.Model.hyper_t / Deriv.Trained.hyper_t
    Evaluation ev;
    Kernel kernel;
    MatP inducing;
  };
  struct Variances_t { Vec variances; double sigma2; };
  struct Covariances_t { MatP points; Mat covariances; double sigma2; std::shared_ptr<Problem> problem; };
  // Mean_predictor.t / Co_variance_predictor.t built from stored numbers (lib/fitc_gp.ml:377-391, :429-447) rather
  // than from a model on the device -- the `test` flow of bin/ocaml_gpr.ml:373-413.  The device problem that serves
  // it is created on first use, sized for the batch it is asked about.
  struct Standalone_t {
    MatP inducing;
    std::optional<Vec> coeffs;
    std::optional<Kernel> kernel;
    std::optional<std::pair<Mat, Mat>> cov_coeffs;  // (chol_km, r_mat)
    std::shared_ptr<std::shared_ptr<Problem>> prob = std::make_shared<std::shared_ptr<Problem>>();
  };
  struct Cov_sampler_t { Vec means; Covariances_t covariances; double add_diag; };

  // one evaluation on the device: multim_f / multim_dcommon, lib/fitc_gp.ml:1601-1636
  static Evaluation run(const Model_t& model, const Vec* targets, bool want_grad, std::shared_ptr<const void> owner) {
    Problem& prob = *model.inputs.problem;
    const Kernel& k = model.inputs.inducing.kernel;
    const Mat& z = *model.inputs.inducing.points;
    if (targets) check(gprhip_set_targets(prob.get(), targets->data()));
    gprhip_hypers h{};
    Spec::fill(k, h);
    h.sigma2 = model.sigma2;
    h.inducing = z.data();
    h.variational = model.variational;
    h.model_only = targets ? 0 : 1;
    h.jitter = cholesky_jitter;
    // Model.update_sigma2 (lib/fitc_gp.ml:234-236): same kernel object and inducing matrix as the problem's
    // previous evaluation -> only sigma2 changed, K_nm / V / r stay on the device
    h.reuse_v = (prob.last_kernel == model.kernel_ref && prob.last_inducing == model.inputs.inducing.points);
    Evaluation ev;
    ev.grad.assign((size_t)std::max<int64_t>(1, gprhip_n_hypers(prob.get(), Spec::flags(k))), 0.0);
    ev.coeffs.assign((size_t)prob.m, 0.0);
    gprhip_result r{};
    check(gprhip_eval(prob.get(), &h, want_grad, &r, ev.grad.data(), ev.coeffs.data()));
    ev.l1 = r.l1; ev.l2 = r.l2; ev.l = r.l; ev.dl_dsigma2 = r.dl_dsigma2;
    ev.has_grad = want_grad;
    if (want_grad) ev.grad.resize((size_t)r.n_hypers);
    prob.last_kernel = model.kernel_ref;
    prob.last_inducing = model.inputs.inducing.points;
    prob.state_owner = std::move(owner);
    return ev;
  }

  template <bool Variational, int CovKind /* 0 FITC, 1 FIC */>
  struct Variant {
    struct Inducing {
      static Inducing_t calc(const Kernel& kernel, MatP points) { return {kernel, std::move(points)}; }
      static MatP get_points(const Inducing_t& i) { return i.points; }
    };
    struct Inputs {
      // Inputs.calc (lib/fitc_gp.ml:108-115).  `train` = these are training inputs: make them device-resident.
      static Inputs_t calc(const Inducing_t& inducing, MatP points, bool train = true, int device = 0,
                           int precision = GPRHIP_F64) {
        const int d = Spec::kernel_space_dim(inducing.kernel, *points);
        if (inducing.points->rows != d)
          throw Failure(GPRHIP_EBADARG, "Inputs.calc: inducing points and kernel space disagree about the dimension");
        Inputs_t in{inducing, points, nullptr};
        if (train) {
          in.problem = std::make_shared<Problem>(Spec::cov_kind, points->cols, points->rows, d,
                                                 inducing.points->cols, device, precision);
          check(gprhip_set_inputs(in.problem->get(), points->data(), points->rows));
        }
        return in;
      }
      // the same training inputs under new hyper-parameters (what the optimiser does every iteration)
      static Inputs_t recalc(const Inputs_t& old, const Inducing_t& inducing) {
        return {inducing, old.points, old.problem};
      }
    };
    struct Model {
      static Model_t calc(const Inputs_t& inputs, double sigma2) {
        if (sigma2 < 0.0) throw Failure(GPRHIP_EBADARG, "Model.check_sigma2: sigma2 < 0");  // lib/fitc_gp.ml:148-149
        if (!inputs.problem) throw Failure(GPRHIP_ESTATE, "Model.calc: inputs were not created as training inputs");
        Model_t m;
        m.inputs = inputs;
        m.sigma2 = sigma2;
        m.variational = Variational;
        m.kernel_ref = std::make_shared<Kernel>(inputs.inducing.kernel);
        return m;
      }
      static Model_t update_sigma2(const Model_t& model, double sigma2) {  // lib/fitc_gp.ml:234-236
        Model_t m = calc(model.inputs, sigma2);
        m.kernel_ref = model.kernel_ref;  // same kernel identity: the device keeps K_nm, V, r
        return m;
      }
      static const Evaluation& evaluation(const Model_t& m, bool want_grad) {
        auto it = m.ev->find(true);
        if (it != m.ev->end()) return it->second;
        it = m.ev->find(want_grad);
        if (it == m.ev->end()) it = m.ev->emplace(want_grad, run(m, nullptr, want_grad, m.id)).first;
        return it->second;
      }
      static void ensure_state(const Model_t& m) {
        if (m.inputs.problem->state_owner == m.id) return;
        (*m.ev)[false] = run(m, nullptr, false, m.id);
      }
      static double calc_log_evidence(const Model_t& m) { return evaluation(m, false).l1; }        // :238
      static double calc_log_evidence_sigma2(const Model_t& m) { return evaluation(m, true).dl_dsigma2; }  // :1121
      static Hyper_t prepare_hyper(const Model_t& m) {                                             // :1126-1136
        return {evaluation(m, true), m.inputs.inducing.kernel, m.inputs.inducing.points};
      }
      static double calc_log_evidence(const Hyper_t& ht, const Hyper& h) {
        return ht.ev.grad[(size_t)Spec::index_of(ht.kernel, *ht.inducing, h)];
      }
      static std::pair<Mat, Mat> calc_co_variance_coeffs(const Model_t& m) {  // :240
        ensure_state(m);
        const int mm = m.inputs.problem->m;
        Mat u(mm, mm), r(mm, mm);
        check(gprhip_co_variance_coeffs(m.inputs.problem->get(), u.data(), r.data()));
        return {u, r};
      }
      static double get_sigma2(const Model_t& m) { return m.sigma2; }
    };
    struct Trained {
      static Trained_t calc(const Model_t& model, const Vec& targets, bool want_grad = true) {
        if ((int64_t)targets.size() != model.inputs.problem->n)  // lib/fitc_gp.ml:283-284
          throw Failure(GPRHIP_EBADARG, "Trained.calc: Vec.dim targets (" + std::to_string(targets.size()) +
                                            ") <> n (" + std::to_string(model.inputs.problem->n) + ")");
        Trained_t t;
        t.model = model;
        t.targets = targets;
        t.want_grad = want_grad;
        return t;
      }
      static const Evaluation& evaluation(const Trained_t& t) {
        if (!*t.ev) *t.ev = run(t.model, &t.targets, t.want_grad, t.id);
        return **t.ev;
      }
      static void ensure_state(const Trained_t& t) {
        if (*t.ev && t.model.inputs.problem->state_owner == t.id) return;
        *t.ev = run(t.model, &t.targets, t.want_grad, t.id);
      }
      static double calc_log_evidence(const Trained_t& t) { return evaluation(t).l; }                 // :295
      static const Vec& calc_mean_coeffs(const Trained_t& t) { return evaluation(t).coeffs; }         // :294
      static double calc_log_evidence_sigma2(const Trained_t& t) { return evaluation(t).dl_dsigma2; } // :1187
      static Hyper_t prepare_hyper(const Trained_t& t) {                                              // :1192-1207
        if (!t.want_grad) throw Failure(GPRHIP_ESTATE, "Trained.prepare_hyper: created without derivatives");
        return {evaluation(t), t.model.inputs.inducing.kernel, t.model.inputs.inducing.points};
      }
      static double calc_log_evidence(const Hyper_t& ht, const Hyper& h) {                            // :1005-1021
        return ht.ev.grad[(size_t)Spec::index_of(ht.kernel, *ht.inducing, h)];
      }
      static Vec calc_means(const Trained_t& t) {                                                     // :296-297
        ensure_state(t);
        Vec means((size_t)t.model.inputs.problem->n);
        double sums[4];
        check(gprhip_train_stats(t.model.inputs.problem->get(), means.data(), sums));
        return means;
      }
    };
    struct Stats {
      static Stats_t calc(const Trained_t& t) {  // lib/fitc_gp.ml:353-373
        Trained::ensure_state(t);
        double s[4];
        check(gprhip_train_stats(t.model.inputs.problem->get(), nullptr, s));
        const double n = (double)t.targets.size();
        Stats_t st;
        st.n_samples = (int64_t)t.targets.size();
        st.target_variance = s[3] / n;
        st.sse = s[0];
        st.mse = s[0] / n;
        st.rmse = std::sqrt(st.mse);
        st.smse = st.mse / st.target_variance;
        const double pi = 3.14159265358979323846;
        st.msll = (-0.5 * std::log(2.0 * pi * st.target_variance) - 0.5) - Trained::evaluation(t).l / n;
        st.mad = s[1] / n;
        st.maxad = s[2];
        return st;
      }
    };
    // ---- prediction at new inputs from a trained model (Means / Variances / Covariances, :416-627)
    static Problem& problem_for(const Trained_t& t, const Inputs_t& in, const char* who) {
      if (in.inducing.points != t.model.inputs.inducing.points)  // phys_equal check, :419-424
        throw Failure(GPRHIP_EBADARG, std::string(who) + ": trained and inputs disagree about inducing points");
      Trained::ensure_state(t);
      return *t.model.inputs.problem;
    }
    static Problem& problem_for(const Model_t& m, const Inputs_t& in, const char* who) {
      if (in.inducing.points != m.inputs.inducing.points)
        throw Failure(GPRHIP_EBADARG, std::string(who) + ": co-variance predictor and inputs disagree about "
                                                           "inducing points");
      Model::ensure_state(m);
      return *m.inputs.problem;
    }
    static Problem& problem_for(const Standalone_t& sa, const Inputs_t& in, const char* who, double sigma2 = 0.0) {
      if (in.inducing.points != sa.inducing)
        throw Failure(GPRHIP_EBADARG, std::string(who) + ": predictor and inputs disagree about inducing points");
      const Kernel& k = sa.kernel ? *sa.kernel : in.inducing.kernel;
      const Mat& pts = *in.points;
      std::shared_ptr<Problem>& prob = *sa.prob;
      if (!prob || prob->D != pts.rows || prob->n < pts.cols)
        prob = std::make_shared<Problem>(Spec::cov_kind, std::max(pts.cols, 1024), pts.rows, sa.inducing->rows,
                                         sa.inducing->cols);
      gprhip_hypers h{};
      Spec::fill(k, h);
      h.sigma2 = sigma2;
      h.inducing = sa.inducing->data();
      h.jitter = cholesky_jitter;
      check(gprhip_load_predictor(prob->get(), &h, sa.coeffs ? sa.coeffs->data() : nullptr,
                                  sa.cov_coeffs ? sa.cov_coeffs->first.data() : nullptr,
                                  sa.cov_coeffs ? sa.cov_coeffs->second.data() : nullptr));
      prob->state_owner = sa.prob;
      return *prob;
    }
    static Problem& problem_of(const Trained_t& t, const Inputs_t& in, const char* who, double) {
      return problem_for(t, in, who);
    }
    static Problem& problem_of(const Model_t& m, const Inputs_t& in, const char* who, double) {
      return problem_for(m, in, who);
    }
    static Problem& problem_of(const Standalone_t& sa, const Inputs_t& in, const char* who, double sigma2) {
      return problem_for(sa, in, who, sigma2);
    }
    static std::shared_ptr<Problem> owner_problem(const Trained_t& t) { return t.model.inputs.problem; }
    static std::shared_ptr<Problem> owner_problem(const Model_t& m) { return m.inputs.problem; }
    static std::shared_ptr<Problem> owner_problem(const Standalone_t& sa) { return *sa.prob; }
    struct Mean_predictor {
      static const Trained_t& calc_trained(const Trained_t& t) { return t; }  // :380-384
      static Standalone_t calc(MatP inducing_points, const Vec& coeffs) {     // :386-391
        if ((size_t)inducing_points->cols != coeffs.size())
          throw Failure(GPRHIP_EBADARG, "Mean_predictor.calc: number of inducing points disagrees with dimension of "
                                        "coefficients");
        Standalone_t sa;
        sa.inducing = std::move(inducing_points);
        sa.coeffs = coeffs;
        return sa;
      }
    };
    struct Co_variance_predictor {
      static const Model_t& calc_model(const Model_t& m) { return m; }        // :438-444
      static Standalone_t calc(const Kernel& kernel, MatP inducing_points, const std::pair<Mat, Mat>& coeffs) {  // :446-447
        Standalone_t sa;
        sa.inducing = std::move(inducing_points);
        sa.kernel = kernel;
        sa.cov_coeffs = coeffs;
        return sa;
      }
    };
    struct Means {
      static Vec calc(const Standalone_t& mean_predictor, const Inputs_t& in) {
        Problem& p = problem_for(mean_predictor, in, "Means.calc");
        Vec means((size_t)in.points->cols);
        check(gprhip_predict(p.get(), in.points->data(), in.points->rows, in.points->cols, 0, means.data(), nullptr));
        return means;
      }
      static Vec calc(const Trained_t& mean_predictor, const Inputs_t& in) {  // :418-425
        Problem& p = problem_for(mean_predictor, in, "Means.calc");
        Vec means((size_t)in.points->cols);
        check(gprhip_predict(p.get(), in.points->data(), in.points->rows, in.points->cols, 0, means.data(), nullptr));
        return means;
      }
    };
    struct Variances {
      template <class Owner>
      static Variances_t calc(const Owner& cvp, double sigma2, const Inputs_t& in) {  // :498-518
        Problem& p = problem_of(cvp, in, "Variances.calc", sigma2);
        Variances_t v{Vec((size_t)in.points->cols), sigma2};
        check(gprhip_predict(p.get(), in.points->data(), in.points->rows, in.points->cols, 0, nullptr,
                             v.variances.data()));
        return v;
      }
      static Vec get(const Variances_t& v, bool predictive = true) {  // :520-529
        Vec out = v.variances;
        if (predictive)
          for (double& x : out) x += v.sigma2;
        return out;
      }
    };
    struct Covariances {  // FITC_covariances / FIC_covariances, :565-627
      template <class Owner>
      static Covariances_t calc(const Owner& cvp, double sigma2, const Inputs_t& in) {
        Problem& p = problem_of(cvp, in, CovKind ? "FIC_covariances.calc" : "FITC_covariances.calc", sigma2);
        Covariances_t c{in.points, Mat(in.points->cols, in.points->cols), sigma2,
                        owner_problem(cvp)};
        check(gprhip_covariances(p.get(), in.points->data(), in.points->rows, in.points->cols, CovKind, 0,
                                 c.covariances.data()));
        return c;
      }
      static Mat get(const Covariances_t& c, bool predictive = true) {  // :549-559
        Mat out = c.covariances;
        if (predictive)
          for (int i = 0; i < out.rows; ++i) out(i, i) += c.sigma2;
        return out;
      }
    };
    struct Cov_sampler {  // Common_cov_sampler, lib/fitc_gp.ml:656-697; the standard normal draws z are the caller's
      static Cov_sampler_t calc(const Vec& means, const Covariances_t& cov, bool predictive = true) {  // :659-675
        if ((int)means.size() != cov.covariances.rows)
          throw Failure(GPRHIP_EBADARG, "Cov_sampler: means and covariances disagree about input points");
        Cov_sampler_t s{means, cov, predictive ? cov.sigma2 : 0.0};
        Mat z0(cov.covariances.rows, 1);
        (void)samples(s, z0);  // factor now: a potrf failure surfaces in calc, as in the reference
        return s;
      }
      static Mat samples(const Cov_sampler_t& s, const Mat& z) {  // :685-697: means + cov_chol^T z, per column
        const int nt = s.covariances.covariances.rows;
        if (z.rows != nt) throw Failure(GPRHIP_EBADARG, "Cov_sampler.samples: z has the wrong number of rows");
        Mat out(nt, z.cols);
        check(gprhip_cov_samples(s.covariances.problem->get(), s.covariances.covariances.data(), nt, nt, s.add_diag,
                                 cholesky_jitter, s.means.data(), z.data(), z.cols, out.data()));
        return out;
      }
    };
    struct Optim {
      // Optim.calc_gradient (lib/fitc_gp.ml:1674-1694): [dl/dsigma2 * sigma2 (if learnt); dl/dhyper ...]
      static Vec calc_gradient(bool learn_sigma2, double sigma2, const std::vector<Hyper>& hypers,
                               const Trained_t& trained) {
        const Hyper_t ht = Trained::prepare_hyper(trained);
        Vec g;
        if (learn_sigma2) g.push_back(Trained::calc_log_evidence_sigma2(trained) * sigma2);
        for (const Hyper& h : hypers) g.push_back(Trained::calc_log_evidence(ht, h));
        return g;
      }
    };
    struct Test {
      // Deriv.Test.self_test (lib/fitc_gp.ml:1398-1462): forward finite difference (eps) of the model and trained
      // log evidence against the analytic derivative; throws like the reference's failwithf.  hyper == nullptr
      // checks sigma2.
      static void self_test(const Kernel& kernel, MatP inducing_points, MatP points, double sigma2,
                            const Vec& targets, const Hyper* hyper, double eps = 1e-8, double tol = 1e-2) {
        auto build = [&](const Kernel& k, MatP z, double s2) {
          auto ind = Inducing::calc(k, z);
          auto inp = Inputs::calc(ind, points);
          auto mod = Model::calc(inp, s2);
          return std::make_pair(mod, Trained::calc(mod, targets, true));
        };
        auto [mod1, tr1] = build(kernel, inducing_points, sigma2);
        double m1 = Model::evaluation(mod1, true).l1, t1 = Trained::evaluation(tr1).l, dm, dt, m2, t2;
        if (!hyper) {
          dm = Model::calc_log_evidence_sigma2(mod1);
          dt = Trained::calc_log_evidence_sigma2(tr1);
          auto [mod2, tr2] = build(kernel, inducing_points, sigma2 + eps);
          m2 = Model::calc_log_evidence(mod2);
          t2 = Trained::calc_log_evidence(tr2);
        } else {
          dm = Model::calc_log_evidence(Model::prepare_hyper(mod1), *hyper);
          dt = Trained::calc_log_evidence(Trained::prepare_hyper(tr1), *hyper);
          const double v = Spec::get_value(kernel, *inducing_points, *hyper);
          auto [k2, z2] = Spec::set_values(kernel, inducing_points, {*hyper}, {v + eps});
          auto [mod2, tr2] = build(k2, z2, sigma2);
          m2 = Model::calc_log_evidence(mod2);
          t2 = Trained::calc_log_evidence(tr2);
        }
        auto bad = [&](double before, double after, double deriv, const char* what) {
          const double fd = (after - before) / eps;
          if (!(std::fabs(fd - deriv) <= tol))  // is_bad_deriv, :1219-1221 (NaN-safe)
            throw Failure(GPRHIP_EBADARG, std::string("Gpr.Fitc_gp.Make_deriv.Test.self_test: finite difference (") +
                                              std::to_string(fd) + ") and derivative (" + std::to_string(deriv) +
                                              ") differ by more than " + std::to_string(tol) + " on " + what);
        };
        bad(m1, m2, dm, "model");
        bad(t1, t2, dt, "trained");
      }
    };
  };

  using FITC = Variant<false, 0>;
  using Variational_FITC = Variant<true, 0>;
  using FIC = Variant<false, 1>;
  using Variational_FIC = Variant<true, 1>;
};

}  // namespace gpr

#endif  // GPRHIP_HPP
