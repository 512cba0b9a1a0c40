"""Host-side mirror of Gpr.Fitc_gp.Make_deriv (reference lib/fitc_gp.mli:83-135, lib/interfaces.ml
Sigs.Deriv :848-1154) for the FITC evidence + gradient path, backed by the HIP library.

    GP = fitc_gp.Make_deriv(cov_se_iso)          # module GP = Fitc_gp.Make_deriv (Cov_se_iso.Deriv)
    FITC = GP.FITC                               # test/save_data.ml:24-27
    inducing = FITC.Deriv.Inducing.calc(kernel, inducing_points)
    inputs   = FITC.Deriv.Inputs.calc(inducing, training_inputs)
    model    = FITC.Deriv.Model.calc(inputs, sigma2=0.1)
    trained  = FITC.Deriv.Trained.calc(model, targets=y)
    l        = FITC.Eval.Trained.calc_log_evidence(FITC.Deriv.Trained.calc_eval(trained))
    hyper_t  = FITC.Deriv.Trained.prepare_hyper(trained)
    dl       = FITC.Deriv.Trained.calc_log_evidence(hyper_t, hyper)

`Inputs.t`, `Model.t`, `Trained.t`, `hyper_t` are abstract in the reference signature
(lib/interfaces.ml:433, :459, :514, :895-899, :944-948); here they are small Python objects
holding a handle to the device-resident problem.  The reference computes its matrices eagerly
stage by stage; the device path is a two-pass streaming evaluation, so these objects are lazy and
the whole evaluation runs when the first number (log evidence, gradient entry) is requested.
Only scalars, m-vectors and the gradient ever cross back to the host.

Beyond the hot path: posterior means and variances at new inputs (`Eval.Mean(s)`,
`Eval.Variance(s)`), training-set statistics (`Eval.Stats`), posterior covariance matrices
(`Eval.Covariances`: FITC_covariances in `FITC`/`Variational_FITC`, FIC_covariances in `FIC`/
`Variational_FIC`, lib/fitc_gp.ml:565-627 -- the only place the two families differ) and the samplers
(`Eval.Sampler`, `Eval.Cov_sampler`; the standard normal draws come from a numpy Generator instead of
GSL's ziggurat).  `Inducing.choose_n_first_inputs` / `choose_n_random_inputs`, `Inputs.create_default_kernel` and the
`calc_model_inputs` variants (n x n host matrices: small n only) complete Sigs.Eval.  The
optimiser drivers live in gpr_amd/optim.py: an L-BFGS driver over the reference's multim_f/multim_dcommon callbacks
(GSL itself is not available) and step-for-step mirrors of Optim.SGD / Optim.SMD.
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np

from .problem import CHOLESKY_JITTER, Problem


class _Inducing:
    def __init__(self, kernel, points):
        self.kernel = kernel
        self.points = np.asfortranarray(points, dtype=np.float64)


class _Inputs:
    """Eval/Deriv Inputs.t: binds points to an inducing set; the device copy is made on first use
    (training inputs) and never for points that are only predicted at."""

    def __init__(self, inducing, points, make_problem):
        self.inducing = inducing
        self.points = points
        self._make_problem = make_problem
        self._problem = None

    @property
    def problem(self):
        # (the functor's cache may have evicted and closed the problem since: make it again)
        if self._problem is None or not self._problem.is_open():
            self._problem = self._make_problem()
        return self._problem


class _Model:
    def __init__(self, inputs, sigma2, variational):
        if sigma2 < 0.0:  # Model.check_sigma2, lib/fitc_gp.ml:148-149
            raise ValueError("Model.check_sigma2: sigma2 < 0")
        self.inputs, self.sigma2, self.variational = inputs, float(sigma2), variational
        self._ev = {}

    def evaluation(self, want_grad):
        if True in self._ev:  # the gradient evaluation also carries the evidence
            return self._ev[True]
        key = bool(want_grad)
        if key not in self._ev:
            self._ev[key] = _run(self, None, key, self)
        return self._ev[key]

    def ensure_state(self):
        """Make the device hold this model's chol_km / r_mat: the problem is shared by every model and
        trained object built on the same inputs, and keeps the state of whichever evaluation ran last."""
        owner = getattr(self.inputs.problem, "_state_owner", None)
        if owner is self or (isinstance(owner, _Trained) and owner.model is self):
            return
        self._ev[False] = _run(self, None, False, self)


class _Trained:
    def __init__(self, model, targets, want_grad):
        self.model = model
        self.targets = np.ascontiguousarray(targets, dtype=np.float64)
        n = model.inputs.problem.n
        if self.targets.shape != (n,):  # lib/fitc_gp.ml:283-284
            raise ValueError("Trained.calc: Vec.dim targets (%d) <> n (%d)" % (self.targets.shape[0], n))
        self.want_grad = want_grad
        self._ev = None

    def evaluation(self):
        if self._ev is None:
            self._ev = _run(self.model, self.targets, self.want_grad, self)
        return self._ev

    def ensure_state(self):
        """Make the device hold this trained model's state (see _Model.ensure_state)."""
        if self._ev is None or getattr(self.model.inputs.problem, "_state_owner", None) is not self:
            self._ev = _run(self.model, self.targets, self.want_grad, self)


class _HyperT:
    def __init__(self, evaluation, kernel, inducing_points, spec):
        self.evaluation, self.kernel, self.inducing_points, self.spec = evaluation, kernel, inducing_points, spec


def _run(model, targets, want_grad, owner):
    inputs = model.inputs
    prob = inputs.problem
    spec = prob._spec
    kernel = inputs.inducing.kernel
    if targets is not None:  # n doubles: cheap next to one evaluation, and never stale
        prob.set_targets(targets)
    # Model.update_sigma2 (lib/fitc_gp.ml:234-236, :1083-1090): when the problem's previous evaluation used
    # this very kernel object and inducing matrix, only sigma2 changed -> K_nm, V, r stay on the device
    sig = (id(kernel), id(inputs.inducing.points))
    reuse = getattr(prob, "_last_sig", None) == sig
    # the device state belongs to nobody while the evaluation runs: if it raises (e.g. NotPositiveDefinite, which the
    # optimiser driver catches and survives), no earlier model may go on predicting from -- or reusing V of -- a
    # half-overwritten state
    prob._last_sig = prob._state_owner = None
    ev = prob.eval(sigma2=model.sigma2, inducing=inputs.inducing.points, variational=model.variational,
                   model_only=targets is None, want_grad=want_grad, jitter=prob._jitter, reuse_v=reuse,
                   **spec.eval_args(kernel))
    prob._last_sig = sig
    prob._state_owner = owner
    prob._last_refs = (kernel, inputs.inducing.points)  # keep the ids alive
    return ev


class _Standalone:
    """Mean_predictor.t / Co_variance_predictor.t built from stored numbers (lib/fitc_gp.ml:377-391, :429-447)
    rather than from a model on the device: the `test` flow of bin/ocaml_gpr.ml:373-413.  The device problem that
    serves it is created on first use, sized for the batch of points it is asked about."""

    def __init__(self, spec, functor, inducing_points, coeffs=None, kernel=None, cov_coeffs=None):
        self.spec, self.functor = spec, functor
        self.inducing_points = inducing_points
        self.coeffs, self.kernel, self.cov_coeffs = coeffs, kernel, cov_coeffs
        self._prob = None
        self._loaded_for = None

    def problem_for(self, kernel, points, sigma2):
        D, nt = points.shape
        d, m = self.inducing_points.shape
        if self._prob is None or self._prob.D != D or self._prob.n < nt:
            if self._prob is not None:
                self._prob.close()
            self._prob = Problem(self.spec.COV_KIND, max(nt, 1024), D, d, m, precision=self.functor.precision)
            self.functor._standalone.append(self._prob)
            self._loaded_for = None
        key = (id(kernel), float(sigma2))
        if self._loaded_for != key:
            self._prob.load_predictor(sigma2=sigma2, inducing=self.inducing_points, coeffs=self.coeffs,
                                      co_variance_coeffs=self.cov_coeffs, jitter=self.functor.jitter,
                                      **self.spec.eval_args(kernel))
            self._loaded_for = key
            self._kernel_ref = kernel
        return self._prob


_default_rng = np.random.default_rng()


def _rng(rng):
    if rng is None:
        return _default_rng
    return rng if isinstance(rng, np.random.Generator) else np.random.default_rng(rng)


def _make_variant(spec, variational, functor, cov_kind="FITC"):
    def inducing_calc(kernel, points):
        return _Inducing(kernel, points)

    def inputs_calc(inducing, points, device=0, chunk_rows=0):
        points = np.asarray(points)
        d = spec.kernel_space_dim(inducing.kernel, points)
        m = inducing.points.shape[1]
        if inducing.points.shape[0] != d:
            raise ValueError("Inputs.calc: inducing points have dimension %d, kernel space has %d"
                             % (inducing.points.shape[0], d))
        key = (id(points), points.shape, m, d, device, chunk_rows)

        def make_problem():
            prob = functor._problems.get(key)
            if prob is None:
                D, n = points.shape
                prob = Problem(spec.COV_KIND, n, D, d, m, device=device, chunk_rows=chunk_rows,
                               precision=functor.precision)
                prob.set_inputs(points)
                prob._spec, prob._jitter = spec, functor.jitter
                prob._points_ref = points  # keep the id() stable
                if len(functor._problems) >= 4:  # resident copies are large; keep a handful
                    functor._problems.pop(next(iter(functor._problems))).close()
                functor._problems[key] = prob
            return prob

        return _Inputs(inducing, points, make_problem)

    def model_calc(inputs, sigma2):
        return _Model(inputs, sigma2, variational)

    def model_update_sigma2(model, sigma2):
        return _Model(model.inputs, sigma2, variational)

    def check_n_inducing(n_inducing, inputs):
        n_inputs = np.asarray(inputs).shape[1]
        if n_inputs < 1 or n_inducing > n_inputs:  # lib/fitc_gp.ml:45-51
            raise ValueError("Gpr.Fitc_gp.Make_common.check_n_inducing: violating 1 <= n_inducing (%d) <= n_inputs (%d)"
                             % (n_inducing, n_inputs))

    def choose(kernel, inputs, indexes):
        """Inducing.choose (lib/fitc_gp.ml:62-64): Utils.choose_cols, then the spec's create_inducing.  Returns the
        inducing POINTS (Spec.Inducing.t, lib/interfaces.ml:382-395), which the caller hands to Inducing.calc."""
        chosen = np.asfortranarray(np.asarray(inputs, dtype=np.float64)[:, indexes])
        return spec.create_inducing(kernel, chosen)

    def choose_n_first_inputs(kernel, inputs, n_inducing):
        check_n_inducing(n_inducing, inputs)                                            # :66-72
        return choose(kernel, inputs, np.arange(n_inducing))

    def choose_n_random_inputs(kernel, inputs, n_inducing, rnd_state=None):
        """lib/fitc_gp.ml:74-92: the first n_inducing steps of a Fisher-Yates shuffle of the input indexes.  The
        reference draws `Random.State.int rnd_state (n_inputs - i + 1)` at step i; a numpy Generator (or a seed) stands
        in for OCaml's Random.State, drawing `integers(n_inputs - i + 1)` at the same step."""
        check_n_inducing(n_inducing, inputs)
        rng = _rng(rnd_state)
        n_inputs = np.asarray(inputs).shape[1]
        indexes = np.arange(n_inputs)
        for i in range(n_inducing):   # (0-based; the reference's step i+1 draws below n_inputs - i)
            rnd_index = int(rng.integers(n_inputs - i))
            indexes[rnd_index], indexes[i] = indexes[i], indexes[rnd_index]
        return choose(kernel, inputs, indexes[:n_inducing])

    def hyper_lookup(hyper_t, hyper):
        idx = spec.HyperModule.index_of(hyper_t.kernel, hyper_t.inducing_points, hyper)
        return float(hyper_t.evaluation.grad[idx])

    Eval = SimpleNamespace(
        Inducing=SimpleNamespace(calc=inducing_calc, get_points=lambda i: i.points,
                                 get_kernel=lambda i: i.kernel, choose_n_first_inputs=choose_n_first_inputs,
                                 choose_n_random_inputs=choose_n_random_inputs),
        Inputs=SimpleNamespace(calc=lambda points, inducing, **kw: inputs_calc(inducing, points, **kw),
                               get_points=lambda i: i.points,
                               create_default_kernel=lambda inputs, n_inducing, **kw: spec.Kernel.create(  # :128-130
                                   spec.create_default_kernel_params(inputs, n_inducing, **kw))),
        Model=SimpleNamespace(
            calc=model_calc, update_sigma2=model_update_sigma2,
            calc_log_evidence=lambda model: model.evaluation(False).l1,   # lib/fitc_gp.ml:238
            get_sigma2=lambda model: model.sigma2, get_inputs=lambda model: model.inputs,
            get_inducing=lambda model: model.inputs.inducing,
            get_kernel=lambda model: model.inputs.inducing.kernel),
        Trained=SimpleNamespace(
            calc=lambda model, targets: _Trained(model, targets, False),   # multim_f, :1601-1610
            calc_log_evidence=lambda trained: trained.evaluation().l,      # :295
            calc_mean_coeffs=lambda trained: trained.evaluation().coeffs,  # :294
            get_model=lambda trained: trained.model, get_targets=lambda trained: trained.targets),
    )

    # ---- prediction (lib/fitc_gp.ml:377-531): means and variances at new inputs, on the device that
    # holds the trained model's state
    def _problem_of(owner, inputs, sigma2=0.0):
        """The device problem holding `owner`'s state, after the reference's phys_equal check on inducing points
        (lib/fitc_gp.ml:419-424, :499-506, :537-546)."""
        if isinstance(owner, _Standalone):
            if inputs.inducing.points is not owner.inducing_points:
                raise ValueError("Means.calc: trained and inputs disagree about inducing points")
            kernel = owner.kernel if owner.kernel is not None else inputs.inducing.kernel
            return owner.problem_for(kernel, np.asarray(inputs.points), sigma2)
        if inputs.inducing.points is not _model_of(owner).inputs.inducing.points:
            raise ValueError("Means.calc: trained and inputs disagree about inducing points")
        owner.ensure_state()
        return _model_of(owner).inputs.problem

    def _predict(model_or_trained, inputs, predictive, want_variances, sigma2=0.0):
        return _problem_of(model_or_trained, inputs, sigma2).predict(inputs.points, predictive=predictive,
                                                                     want_variances=want_variances)

    def _model_of(obj):
        return obj.model if isinstance(obj, _Trained) else obj

    def mean_predictor_calc(inducing_points, coeffs):
        if inducing_points.shape[1] != np.asarray(coeffs).shape[0]:                        # :387-390
            raise ValueError("Mean_predictor.calc: number of inducing points disagrees with dimension of "
                             "coefficients")
        return _Standalone(spec, functor, inducing_points, coeffs=np.asarray(coeffs, dtype=np.float64))

    Eval.Mean_predictor = SimpleNamespace(
        calc_trained=lambda trained: trained,                                              # :380-384
        calc=mean_predictor_calc,                                                          # :386-391
        get_inducing=lambda mp: mp.inducing_points if isinstance(mp, _Standalone)
        else mp.model.inputs.inducing.points,
        get_coeffs=lambda mp: mp.coeffs if isinstance(mp, _Standalone) else mp.evaluation().coeffs)
    Eval.Means = SimpleNamespace(
        calc=lambda mean_predictor, inputs: _predict(mean_predictor, inputs, False, False)[0],  # :418-425
        get=lambda means: means)
    Eval.Co_variance_predictor = SimpleNamespace(
        calc_model=lambda model: model,                                                    # :438-444
        calc=lambda kernel, inducing_points, co_variance_coeffs:                           # :446-447
        _Standalone(spec, functor, inducing_points, kernel=kernel, cov_coeffs=co_variance_coeffs))

    def calc_co_variance_coeffs(model):                                                    # :240
        model.ensure_state()
        return model.inputs.problem.co_variance_coeffs()

    Eval.Model.calc_co_variance_coeffs = calc_co_variance_coeffs

    class _Variances:
        def __init__(self, variances, sigma2):
            self.variances, self.sigma2 = variances, sigma2

    def variances_calc(cvp, sigma2, inputs):
        # the state (chol_km, r_mat) comes from an evaluation of `cvp` (a model or a trained model)
        return _Variances(_predict(cvp, inputs, False, True, sigma2)[1], sigma2)

    def variances_calc_model_inputs(model):
        """Variances.calc_model_inputs (lib/fitc_gp.ml:487-496): r_vec + rowsum((K_nm R^-1)^2) at the model's own
        inputs -- the same numbers Variances.calc gives there, so the prediction path serves it."""
        return variances_calc(model, model.sigma2, model.inputs)

    Eval.Variances = SimpleNamespace(
        calc_model_inputs=variances_calc_model_inputs,
        calc=variances_calc,                                                                   # :498-518
        get=lambda v, predictive=True: v.variances + v.sigma2 if predictive else v.variances)  # :520-529

    # ---- single-point prediction (Input / Mean / Variance, lib/fitc_gp.ml:88-101, :402-414, :447-482)
    class _Input:
        def __init__(self, inducing, point):
            self.inducing = inducing
            self.point = np.ascontiguousarray(point, dtype=np.float64)
            self.points = np.asfortranarray(self.point.reshape(-1, 1))

    class _Mean:
        def __init__(self, point, value):
            self.point, self.value = point, value

    class _Variance:
        def __init__(self, point, variance, sigma2):
            self.point, self.variance, self.sigma2 = point, variance, sigma2

    Eval.Input = SimpleNamespace(calc=lambda inducing, point: _Input(inducing, point))
    Eval.Mean = SimpleNamespace(
        calc=lambda mean_predictor, inp: _Mean(inp.point, float(_predict(mean_predictor, inp, False, False)[0][0])),
        get=lambda mean: mean.value)
    Eval.Variance = SimpleNamespace(
        calc=lambda cvp, sigma2, inp: _Variance(inp.point, float(_predict(cvp, inp, False, True, sigma2)[1][0]),
                                                sigma2),
        get=lambda v, predictive=True: v.variance + v.sigma2 if predictive else v.variance)

    # ---- Stats (lib/fitc_gp.ml:304-374): residual sums on the device, the derived figures here
    def stats_calc(trained):
        trained.ensure_state()
        ev = trained.evaluation()
        sums, _ = trained.model.inputs.problem.train_stats()
        n = trained.targets.shape[0]
        sse, sad, maxad, sy2 = (float(x) for x in sums)
        target_variance = sy2 / n                                 # :318
        mse = sse / n
        prior_l = -0.5 * np.log(2.0 * np.pi * target_variance) - 0.5   # :329-330
        return SimpleNamespace(n_samples=n, target_variance=target_variance, sse=sse, mse=mse,
                               rmse=float(np.sqrt(mse)), smse=mse / target_variance,
                               msll=float(prior_l - ev.l / n), mad=sad / n, maxad=maxad)

    Eval.Stats = SimpleNamespace(
        calc=stats_calc,
        calc_n_samples=lambda trained: trained.targets.shape[0],
        calc_target_variance=lambda trained: float(trained.targets @ trained.targets) / trained.targets.shape[0],
        calc_sse=lambda trained: stats_calc(trained).sse, calc_mse=lambda trained: stats_calc(trained).mse,
        calc_rmse=lambda trained: stats_calc(trained).rmse, calc_smse=lambda trained: stats_calc(trained).smse,
        calc_msll=lambda trained: stats_calc(trained).msll, calc_mad=lambda trained: stats_calc(trained).mad,
        calc_maxad=lambda trained: stats_calc(trained).maxad)
    Eval.Trained.calc_means = lambda trained: (trained.ensure_state(),
                                               trained.model.inputs.problem.train_stats(True)[1])[1]  # :296-297

    # ---- Covariances (FITC_covariances / FIC_covariances, lib/fitc_gp.ml:565-627) and samplers (:629-697)
    class _Covariances:
        def __init__(self, points, covariances, sigma2, problem):
            self.points, self.covariances, self.sigma2, self._problem = points, covariances, sigma2, problem

    def covariances_calc(cvp, sigma2, inputs):
        try:
            prob = _problem_of(cvp, inputs, sigma2)
        except ValueError:
            raise ValueError("%s_covariances.calc: co-variance predictor and inputs disagree about "
                             "inducing points" % cov_kind) from None                # :537-546
        return _Covariances(inputs.points, prob.covariances(inputs.points, kind=cov_kind, predictive=False), sigma2, prob)

    def covariances_get(c, predictive=True):
        if not predictive:
            return c.covariances
        res = c.covariances.copy()
        res[np.diag_indices(res.shape[0])] += c.sigma2                          # :549-559
        return res

    def covariances_calc_model_inputs(model):
        """FITC_covariances.calc_model_inputs (lib/fitc_gp.ml:569-579): K_n - V V^T + Q_n Q_n^T, and
        FIC_covariances.calc_model_inputs (:609-614): Q_n Q_n^T + diag(r_vec) -- as written, with the model's own Q
        factor, whose rows carry sqrt(1/s_i) (Q_n = diag(sqrt is) K_nm R^-1, :176-182).  Assembled on the host from
        what the device gives at the model's own inputs: C_fitc = K_n - V V^T + Q' Q'^T (`calc`), Q' Q'^T off the
        diagonal from the FIC form, its diagonal from the variances, r_vec from the model state.
        An n x n host matrix, as in the reference: for small n only."""
        prob = _problem_of(model, model.inputs, model.sigma2)
        pts = model.inputs.points
        r_vec = prob.debug_fetch("r")
        qq = prob.covariances(pts, kind="FIC", predictive=False)
        var = prob.predict(pts, predictive=False, want_variances=True)[1]
        qq[np.diag_indices(qq.shape[0])] = var - r_vec                  # |Q'_i|^2 = variance_i - r_i (:487-496)
        sq = np.sqrt(1.0 / (r_vec + model.sigma2))
        qnqn = qq * sq[:, None] * sq[None, :]
        if cov_kind == "FIC":
            cov = qnqn
            cov[np.diag_indices(cov.shape[0])] += r_vec
        else:
            cov = prob.covariances(pts, kind="FITC", predictive=False) + (qnqn - qq)
        return _Covariances(pts, cov, model.sigma2, prob)

    Eval.Covariances = SimpleNamespace(
        calc=covariances_calc, get=covariances_get, calc_model_inputs=covariances_calc_model_inputs,
        get_variances=lambda c: _Variances(np.diag(c.covariances).copy(), c.sigma2))   # :564-565

    class _Sampler:
        def __init__(self, mean, stddev):
            self.mean, self.stddev = mean, stddev

    def sampler_calc(mean, variance, predictive=True):
        if mean.point is not variance.point:                                    # :633-635
            raise ValueError("%s.Sampler: mean and variance disagree about input point" % cov_kind)
        used = variance.variance + variance.sigma2 if predictive else variance.variance
        return _Sampler(mean.value, float(np.sqrt(used)))

    Eval.Sampler = SimpleNamespace(
        calc=sampler_calc,
        sample=lambda sampler, rng=None: sampler.mean + sampler.stddev * _rng(rng).standard_normal(),
        samples=lambda sampler, n, rng=None: sampler.mean + sampler.stddev * _rng(rng).standard_normal(n))

    class _CovSampler:
        def __init__(self, means, covariances, add_diag):
            self.means, self.covariances, self.add_diag = means, covariances, add_diag

        def draw(self, z):
            return self.covariances._problem.cov_samples(self.covariances.covariances, self.means, z,
                                                         add_diag=self.add_diag, jitter=functor.jitter)

    def cov_sampler_calc(means, covariances, predictive=True, points=None):
        """Common_cov_sampler.calc (:659-675).  `means` is the vector Means.get returns; pass `points` (the
        matrix the means were computed at) to have the reference's phys_equal check applied."""
        if points is not None and points is not covariances.points:
            raise ValueError("%s.Cov_sampler: means and covariances disagree about input points" % cov_kind)
        smp = _CovSampler(np.asarray(means, dtype=np.float64), covariances,
                          covariances.sigma2 if predictive else 0.0)
        smp.draw(np.zeros((smp.means.shape[0], 1)))   # factor now: potrf failures surface in calc, as in :673
        return smp

    Eval.Cov_sampler = SimpleNamespace(
        calc=cov_sampler_calc,
        sample=lambda smp, rng=None: smp.draw(_rng(rng).standard_normal((smp.means.shape[0], 1)))[:, 0],
        samples=lambda smp, n, rng=None: smp.draw(_rng(rng).standard_normal((smp.means.shape[0], n))),
        samples_from=lambda smp, z: smp.draw(z))

    def prepare_hyper_model(model):
        return _HyperT(model.evaluation(True), model.inputs.inducing.kernel, model.inputs.inducing.points, spec)

    def prepare_hyper_trained(trained):
        inducing = trained.model.inputs.inducing
        return _HyperT(trained.evaluation(), inducing.kernel, inducing.points, spec)

    def self_test(kernel, inducing_points, points, sigma2, targets, hyper, eps=1e-8, tol=1e-2):
        """Deriv.Test.self_test (lib/fitc_gp.ml:1398-1462): forward finite difference of the model and
        trained log evidence against the analytic derivative; raises like the reference's failwithf."""
        def evals(k, z, s2):
            ind = inducing_calc(k, z)
            inp = inputs_calc(ind, points)
            mod = model_calc(inp, s2)
            return mod, _Trained(mod, targets, True)

        mod1, tr1 = evals(kernel, inducing_points, sigma2)
        if hyper == "Sigma2":
            mod2, tr2 = evals(kernel, inducing_points, sigma2 + eps)
            checks = [("sigma2(model)", mod1.evaluation(True).l1, mod2.evaluation(False).l1,
                       mod1.evaluation(True).dl_dsigma2),
                      ("sigma2(trained)", tr1.evaluation().l, tr2.evaluation().l, tr1.evaluation().dl_dsigma2)]
        else:
            value = spec.HyperModule.get_value(kernel, inducing_points, points, hyper)
            k2, z2, _ = spec.HyperModule.set_values(kernel, inducing_points, points, [hyper], [value + eps])
            mod2, tr2 = evals(k2, z2, sigma2)
            checks = [("hyper(model)", mod1.evaluation(True).l1, mod2.evaluation(False).l1,
                       hyper_lookup(prepare_hyper_model(mod1), hyper)),
                      ("hyper(trained)", tr1.evaluation().l, tr2.evaluation().l,
                       hyper_lookup(prepare_hyper_trained(tr1), hyper))]
        for name, before, after, deriv in checks:
            finite_el = (after - before) / eps
            if not (abs(finite_el - deriv) <= tol):  # is_bad_deriv, :1219-1221 (NaN-safe)
                raise AssertionError(
                    "Gpr.Fitc_gp.Make_deriv.Test.self_test: finite difference (%f) and derivative (%f) "
                    "differ by more than %f on %s" % (finite_el, deriv, tol, name))

    def calc_gradient(learn_sigma2, sigma2, hypers, trained):
        """Optim.calc_gradient (lib/fitc_gp.ml:1674-1694)."""
        ev = trained.evaluation()
        ht = prepare_hyper_trained(trained)
        g = [hyper_lookup(ht, h) for h in hypers]
        if learn_sigma2:
            g = [ev.dl_dsigma2 * sigma2] + g
        return np.array(g)

    Deriv = SimpleNamespace(
        Spec=spec,
        Inducing=SimpleNamespace(calc=inducing_calc, calc_eval=lambda i: i),
        Inputs=SimpleNamespace(calc=inputs_calc, calc_eval=lambda i: i),
        Model=SimpleNamespace(
            calc=model_calc, update_sigma2=model_update_sigma2, calc_eval=lambda m: m,
            calc_log_evidence_sigma2=lambda model: model.evaluation(True).dl_dsigma2,  # :1121-1122
            prepare_hyper=prepare_hyper_model, calc_log_evidence=hyper_lookup),
        Trained=SimpleNamespace(
            calc=lambda model, targets: _Trained(model, targets, True),                 # :1158-1181
            calc_eval=lambda t: t,
            calc_log_evidence_sigma2=lambda trained: trained.evaluation().dl_dsigma2,   # :1187-1188
            prepare_hyper=prepare_hyper_trained, calc_log_evidence=hyper_lookup),
        Test=SimpleNamespace(self_test=self_test),
        Optim=SimpleNamespace(calc_gradient=calc_gradient),
    )
    return SimpleNamespace(Eval=Eval, Deriv=Deriv)


class Make_deriv:
    """Fitc_gp.Make_deriv (lib/fitc_gp.mli:120-135).  `spec` is gpr_amd.cov_se_iso or gpr_amd.cov_se_fat."""

    def __init__(self, spec, jitter=CHOLESKY_JITTER, precision=0):
        self.spec = spec
        self.jitter = jitter  # read once at functor application, like lib/fitc_gp.ml:33
        self.precision = precision  # gpr_amd.F64 (reference parity) or gpr_amd.F32_BULK
        self._problems = {}
        self._standalone = []
        self.FITC = _make_variant(spec, False, self)
        self.Variational_FITC = _make_variant(spec, True, self)
        self.FIC = _make_variant(spec, False, self, "FIC")
        self.Variational_FIC = _make_variant(spec, True, self, "FIC")

    def close(self):
        for p in list(self._problems.values()) + self._standalone:
            p.close()
        self._problems.clear()
        self._standalone.clear()
