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
