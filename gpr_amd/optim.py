"""Evidence maximisation driver: the step either side of the hot path.

The reference drives `multim_f` / `multim_dcommon` from GSL's BFGS2 (`Optim.Gsl.train`,
lib/fitc_gp.ml:1532-1671).  GSL is not part of this build; the same objective/gradient callbacks
are handed to scipy's L-BFGS here.  The parameter vector is the reference's:
[log sigma2; hyper values in Hyper.get_all order] (lib/fitc_gp.ml:1545-1553), with the sigma2 entry of the
gradient scaled by sigma2 because the optimiser works in log sigma2 (lib/fitc_gp.ml:1549, :1622).
"""
from __future__ import annotations

import numpy as np

from ._lib import NotPositiveDefinite


def train(functor_variant, spec, kernel, inducing, inputs, targets, sigma2=None, learn_sigma2=True,
          hypers=None, max_iter=100, tol=1e-6, report=None):
    """Returns (kernel, inducing, sigma2, log_evidence, n_evaluations) at the best point found.

    functor_variant: e.g. fitc_gp.Make_deriv(cov_se_iso).FITC ; hypers: subset to optimise
    (the reference's ?hypers, lib/fitc_gp.ml:1535), default Hyper.get_all."""
    from scipy.optimize import minimize

    F = functor_variant
    H = spec.HyperModule
    targets = np.ascontiguousarray(targets, dtype=np.float64)
    if sigma2 is None:  # Optim.get_sigma2, lib/fitc_gp.ml:1468-1469
        sigma2 = float(targets @ targets) / targets.shape[0]
    if hypers is None:
        hypers = H.get_all(kernel, inducing, inputs)
    x0 = [H.get_value(kernel, inducing, inputs, h) for h in hypers]
    x0 = np.array(([np.log(sigma2)] if learn_sigma2 else []) + x0)
    state = dict(best=None, n=0)

    def unpack(x):
        s2 = float(np.exp(x[0])) if learn_sigma2 else sigma2
        vals = x[1:] if learn_sigma2 else x
        k, z, _ = H.set_values(kernel, inducing, inputs, hypers, vals)
        return k, z, s2

    def fdf(x):  # multim_fdf, lib/fitc_gp.ml:1641-1647
        state["n"] += 1
        try:
            k, z, s2 = unpack(x)
            ind = F.Deriv.Inducing.calc(k, z)
            model = F.Deriv.Model.calc(F.Deriv.Inputs.calc(ind, inputs), sigma2=s2)
            trained = F.Deriv.Trained.calc(model, targets=targets)
            le = F.Eval.Trained.calc_log_evidence(trained)
            g = F.Deriv.Optim.calc_gradient(learn_sigma2, s2, hypers, trained)
            ok = np.isfinite(le) and np.all(np.isfinite(g))
        except (NotPositiveDefinite, OverflowError):
            ok = False
        if not ok:
            # The reference aborts the whole optimisation here (check_exception, lib/fitc_gp.ml:1523-1528: NaN
            # evidence raises Optim_exception).  A line search that overshoots into a region where K_m or B is
            # numerically singular is routine, so this driver rejects the trial point instead: an objective worse
            # than anything seen, with a zero gradient, makes the line search back off.
            if state["best"] is None:
                raise FloatingPointError("Optim: log evidence is not finite at the starting point")
            state["rejected"] = state.get("rejected", 0) + 1
            return state["worst"] + 1e3 * (1.0 + abs(state["worst"])), np.zeros_like(x)
        state["worst"] = max(state.get("worst", -le), -le)
        if state["best"] is None or le > state["best"][0]:
            state["best"] = (le, k, z, s2)
            if report is not None:
                report(state["n"], le, float(np.linalg.norm(g)))
        return -le, -g

    minimize(fdf, x0, jac=True, method="L-BFGS-B", options=dict(maxiter=max_iter, gtol=tol))
    le, k, z, s2 = state["best"]
    return k, z, s2, le, state["n"]


# ---------------------------------------------------------------------------
# Optim.SGD / Optim.SMD (lib/fitc_gp.ml:1696-2017): the reference's own gradient-ascent drivers over the same
# evidence + gradient evaluation.  Deterministic given the starting point, so they are mirrored step for step.
# ---------------------------------------------------------------------------
def _get_sigma2(targets, sigma2):
    if sigma2 is None:  # Optim.get_sigma2, lib/fitc_gp.ml:1468-1472
        return float(targets @ targets) / targets.shape[0]
    if sigma2 < 0.0:
        raise ValueError("Optim.get_sigma2: sigma2 < 0: %f" % sigma2)
    return float(sigma2)


def _trained_at(F, kernel, inducing, inputs, sigma2, targets):
    ind = F.Deriv.Inducing.calc(kernel, inducing)
    model = F.Deriv.Model.calc(F.Deriv.Inputs.calc(ind, inputs), sigma2=sigma2)
    return F.Deriv.Trained.calc(model, targets=targets)


def _make_test(step, epsabs=0.1, max_iter=None, report=None):
    """make_test, lib/fitc_gp.ml:1696-1722: iterate `step`, keep the state with the best log evidence."""
    def test(t):
        if max_iter is not None and max_iter < 0:
            raise ValueError("Optim.SMD.test: max_iter < 0")
        n = -1 if max_iter is None else max_iter
        best, best_le = t, t.log_evidence()
        while n != 0 and not t.gradient_norm < epsabs:
            t = step(t)
            le = t.log_evidence()
            if le > best_le:
                if report is not None:
                    report(t)
                best, best_le = t, le
            n -= 1
        return best
    return test


class _OptimState:
    def log_evidence(self):
        return self.F.Eval.Trained.calc_log_evidence(self.trained)

    def get_trained(self):
        return self.trained


class SGD(_OptimState):
    """Optim.SGD (lib/fitc_gp.ml:1724-1840): theta += eta * gradient, eta decays by tau / (tau + step)."""

    @staticmethod
    def create(F, spec, kernel, inducing, inputs, targets, tau=100.0, eta0=1e-3, step=0, sigma2=None,
               learn_sigma2=True, hypers=None):
        loc = "Gpr.Fitc_gp.Optim.SGD.create"
        if tau <= 0.0:
            raise ValueError("%s: tau (%f) <= 0" % (loc, tau))
        if eta0 <= 0.0:
            raise ValueError("%s: eta0 (%f) <= 0" % (loc, eta0))
        if step < 0:
            raise ValueError("%s: step (%d) < 0" % (loc, step))
        t = SGD()
        t.F, t.spec, t.inputs = F, spec, inputs
        t.targets = np.ascontiguousarray(targets, dtype=np.float64)
        t.learn_sigma2, t.tau, t.eta, t.step_no = bool(learn_sigma2), float(tau), float(eta0), int(step)
        t.sigma2 = _get_sigma2(t.targets, sigma2)
        t.kernel, t.inducing = kernel, inducing
        H = spec.HyperModule
        t.hypers = H.get_all(kernel, inducing, inputs) if hypers is None else hypers
        t.hyper_vals = np.array([H.get_value(kernel, inducing, inputs, h) for h in t.hypers])
        t.trained = _trained_at(F, kernel, inducing, inputs, t.sigma2, t.targets)
        t.gradient = F.Deriv.Optim.calc_gradient(t.learn_sigma2, t.sigma2, t.hypers, t.trained)
        t.gradient_norm = float(np.linalg.norm(t.gradient))
        return t

    @staticmethod
    def step(t):
        n = SGD()
        n.__dict__.update(t.__dict__)
        ofs = 1 if t.learn_sigma2 else 0
        n.sigma2 = float(np.exp(np.log(t.sigma2) + t.eta * t.gradient[0])) if t.learn_sigma2 else t.sigma2
        n.hyper_vals = t.hyper_vals + t.eta * t.gradient[ofs:]
        n.kernel, n.inducing, _ = t.spec.HyperModule.set_values(t.kernel, t.inducing, t.inputs, t.hypers, n.hyper_vals)
        n.trained = _trained_at(t.F, n.kernel, n.inducing, t.inputs, n.sigma2, t.targets)
        n.gradient = t.F.Deriv.Optim.calc_gradient(t.learn_sigma2, n.sigma2, t.hypers, n.trained)
        n.gradient_norm = float(np.linalg.norm(n.gradient))
        n.eta = t.tau / (t.tau + t.step_no) * t.eta
        n.step_no = t.step_no + 1
        return n

    @staticmethod
    def test(t, epsabs=0.1, max_iter=None, report=None):
        return _make_test(SGD.step, epsabs, max_iter, report)(t)


class SMD(_OptimState):
    """Optim.SMD (lib/fitc_gp.ml:1842-2017): stochastic meta-descent -- per-parameter step sizes eta adapted through
    nu, with the Hessian-vector product approximated by a central difference of gradients (two extra evaluations
    per step).  The index conventions of the reference's update are kept as written (:1995-1998: the hyper update
    reads eta from its first entry while the gradient is read past the sigma2 slot)."""

    @staticmethod
    def create(F, spec, kernel, inducing, inputs, targets, eps=1e-8, lam=None, mu=None, eta0=None, nu0=None,
               sigma2=None, learn_sigma2=True, hypers=None):
        loc = "Gpr.Fitc_gp.Optim.SMD.create"
        lam = 0.1 if lam is None else lam
        if lam < 0.0 or lam > 1.0:
            raise ValueError("%s: violating 0 <= lambda(%f) <= 1" % (loc, lam))
        mu = 1e-3 if mu is None else mu
        if mu < 0.0:
            raise ValueError("%s: violating 0 <= mu(%f)" % (loc, mu))
        t = SMD()
        t.F, t.spec, t.inputs = F, spec, inputs
        t.targets = np.ascontiguousarray(targets, dtype=np.float64)
        t.learn_sigma2, t.eps, t.lam, t.mu = bool(learn_sigma2), float(eps), float(lam), float(mu)
        t.sigma2 = _get_sigma2(t.targets, sigma2)
        t.kernel, t.inducing = kernel, inducing
        H = spec.HyperModule
        t.hypers = H.get_all(kernel, inducing, inputs) if hypers is None else hypers
        t.hyper_vals = np.array([H.get_value(kernel, inducing, inputs, h) for h in t.hypers])
        n_all = len(t.hypers) + (1 if t.learn_sigma2 else 0)
        if eta0 is None:
            t.eta = np.full(n_all, 1e-3)
        else:
            t.eta = np.array(eta0, dtype=np.float64)
            if t.eta.shape[0] != n_all:
                raise ValueError("%s: dim(eta0) = %d <> n_all_hypers(%d)" % (loc, t.eta.shape[0], n_all))
            if np.any(t.eta <= 0.0):
                i = int(np.argmax(t.eta <= 0.0))
                raise ValueError("%s: eta0.{%d} < 0: %f" % (loc, i + 1, t.eta[i]))
        if nu0 is None:
            t.nu = np.full(n_all, 1e-3)
        else:
            t.nu = np.array(nu0, dtype=np.float64)
            if t.nu.shape[0] != n_all:
                raise ValueError("%s: dim(nu0) = %d <> n_all_hypers(%d)" % (loc, t.nu.shape[0], n_all))
        t.trained = _trained_at(F, kernel, inducing, inputs, t.sigma2, t.targets)
        t.gradient = F.Deriv.Optim.calc_gradient(t.learn_sigma2, t.sigma2, t.hypers, t.trained)
        t.gradient_norm = float(np.linalg.norm(t.gradient))
        return t

    @staticmethod
    def step(t):
        n = SMD()
        n.__dict__.update(t.__dict__)
        H = t.spec.HyperModule
        nh = len(t.hypers)
        ofs = 1 if t.learn_sigma2 else 0
        log_s2 = np.log(t.sigma2)

        def grad_at(eps):
            s2 = float(np.exp(log_s2 + eps * t.nu[0])) if t.learn_sigma2 else t.sigma2
            vals = t.hyper_vals + eps * t.nu[ofs:ofs + nh]
            k, z, _ = H.set_values(t.kernel, t.inducing, t.inputs, t.hypers, vals)
            tr = _trained_at(t.F, k, z, t.inputs, s2, t.targets)
            return t.F.Deriv.Optim.calc_gradient(t.learn_sigma2, s2, t.hypers, tr)

        lambda_hessian_nu = (t.lam / (2.0 * t.eps)) * (grad_at(t.eps) - grad_at(-t.eps))
        n.eta = t.eta * np.maximum(0.5, 1.0 + t.mu * t.gradient * t.nu)
        n.sigma2 = float(np.exp(log_s2 + n.eta[0] * t.gradient[0])) if t.learn_sigma2 else t.sigma2
        n.hyper_vals = t.hyper_vals + n.eta[:nh] * t.gradient[ofs:ofs + nh]   # Vec.mul ~n eta ~ofsy old_gradient
        n.nu = t.eta * (t.gradient + lambda_hessian_nu) + t.lam * t.nu
        n.kernel, n.inducing, _ = H.set_values(t.kernel, t.inducing, t.inputs, t.hypers, n.hyper_vals)
        n.trained = _trained_at(t.F, n.kernel, n.inducing, t.inputs, n.sigma2, t.targets)
        n.gradient = t.F.Deriv.Optim.calc_gradient(t.learn_sigma2, n.sigma2, t.hypers, n.trained)
        n.gradient_norm = float(np.linalg.norm(n.gradient))
        return n

    @staticmethod
    def test(t, epsabs=0.1, max_iter=None, report=None):
        return _make_test(SMD.step, epsabs, max_iter, report)(t)
