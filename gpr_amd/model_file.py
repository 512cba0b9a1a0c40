"""The steps either side of the evidence path in the reference's command-line tool (bin/ocaml_gpr.ml): reading
comma-separated samples, standardising them, training, writing/reading a model file and predicting from it.

The reference marshals an OCaml record (bin/ocaml_gpr.ml:179-188, :207-232) -- not portable across OCaml versions,
let alone languages.  Here the same fields go into one `.npz` archive (numpy's documented zip-of-.npy format):

    format            "gprhip-model-1"
    cov               "se_fat"
    sigma2            float                         Model.sigma2
    target_mean       float
    input_means       (D,)                          per input dimension
    input_stddevs     (D,)                          sqrt(sum (x - mean)^2)  -- see standardize_inputs
    d, log_sf2        Cov_se_fat.Params             lib/cov_se_fat.ml:27-36
    tproj             (D, d)            optional
    log_hetero_skedasticity (m,)       optional
    log_multiscales_m05     (d, m)     optional
    inducing_points   (d, m)
    coeffs            (m,)                          Mean_predictor coefficients
    chol_km, r_mat    (m, m) upper                  Model.calc_co_variance_coeffs, lib/fitc_gp.ml:240

Every matrix is stored in the reference's orientation (one point per column).
"""
from __future__ import annotations

import io
from types import SimpleNamespace

import numpy as np

from . import cov_se_fat, fitc_gp, optim

FORMAT = "gprhip-model-1"


def read_samples(source):
    """bin/ocaml_gpr.ml:148-172: one sample per line, comma-separated floats, all lines of equal length.
    `source`: path, file object or a string holding the text."""
    if isinstance(source, str) and "\n" not in source and "," not in source:
        with open(source) as f:
            text = f.read()
    elif isinstance(source, str):
        text = source
    else:
        text = source.read()
    rows = []
    for i, line in enumerate(ln for ln in text.splitlines() if ln.strip()):
        try:
            vals = [float(v) for v in line.split(",")]
        except ValueError as exc:
            raise ValueError("failure '%s' converting sample" % line) from exc
        if rows and len(vals) != len(rows[0]):
            raise ValueError("incompatible dimension of sample in line %d: %s" % (i, line))
        rows.append(vals)
    if not rows:
        raise ValueError("no data")
    return np.array(rows, dtype=np.float64)


def read_training_samples(source):
    """bin/ocaml_gpr.ml:190-201: the last column is the target.  Returns (inputs D x n, targets n)."""
    s = read_samples(source)
    return np.asfortranarray(s[:, :-1].T), s[:, -1].copy()


def standardize_inputs(inputs):
    """bin/ocaml_gpr.ml:254-265: per input dimension, mean and `sqrt (Vec.ssqr ~c:mean input)` -- the root of the
    *sum* of squared deviations, not divided by n, exactly as the reference computes its "stddev".
    Returns (standardised inputs, input_means, input_stddevs)."""
    x = np.asarray(inputs, dtype=np.float64)
    means = x.sum(axis=1) / x.shape[1]
    stddevs = np.sqrt(((x - means[:, None]) ** 2).sum(axis=1))
    return np.asfortranarray((x - means[:, None]) / stddevs[:, None]), means, stddevs


def apply_standardization(inputs, input_means, input_stddevs):
    """bin/ocaml_gpr.ml:390-396."""
    x = np.asarray(inputs, dtype=np.float64)
    return np.asfortranarray((x - input_means[:, None]) / input_stddevs[:, None])


def default_params(big_dim, n_inducing, amplitude=1.0, dim_red=None, log_het_sked=None, multiscale=False, rng=None):
    """bin/ocaml_gpr.ml:268-299: log_sf2 = 2 log amplitude; an optional random projection scaled by 1/big_dim
    (Mat.random: uniform on [-1, 1)); optional constant log_hetero_skedasticity; optional zero multiscales."""
    rng = np.random.default_rng() if rng is None else rng
    d, tproj = big_dim, None
    if dim_red is not None:
        d = min(big_dim, dim_red)
        tproj = rng.uniform(-1.0, 1.0, size=(big_dim, d)) / big_dim
    het = None if log_het_sked is None else np.full(n_inducing, float(log_het_sked))
    ms = np.zeros((d, n_inducing)) if multiscale else None
    return cov_se_fat.Params.create(d, 2.0 * np.log(amplitude), tproj, het, ms)


def train(inputs, targets, n_inducing=10, sigma2=1.0, amplitude=1.0, dim_red=None, log_het_sked=None,
          multiscale=False, max_iter=None, rng=None, functor=None, verbose=False):
    """The `train` command (bin/ocaml_gpr.ml:236-349): centre the targets, standardise the inputs, build the
    Cov_se_fat kernel, pick `n_inducing` random training points as inducing inputs (Optim.Gsl.train's
    ~n_rand_inducing, lib/fitc_gp.ml:1556-1561) and maximise the Variational_FIC evidence over all
    hyper-parameters and log sigma2.  The optimiser is gpr_amd.optim (L-BFGS) in place of GSL's BFGS2.
    Returns a model namespace ready for `save_model` / `predict`."""
    rng = np.random.default_rng() if rng is None else rng
    targets = np.asarray(targets, dtype=np.float64)
    target_mean = float(targets.sum() / targets.shape[0])
    y = targets - target_mean
    x, input_means, input_stddevs = standardize_inputs(inputs)
    big_dim, n = x.shape
    m = min(n_inducing, n)
    params = default_params(big_dim, m, amplitude, dim_red, log_het_sked, multiscale, rng)
    kernel = cov_se_fat.Kernel.create(params)
    GP = functor if functor is not None else fitc_gp.Make_deriv(cov_se_fat)
    F = GP.Variational_FIC
    chosen = np.sort(rng.permutation(n)[:m])
    z0 = x[:, chosen]
    if params.tproj is not None:  # Spec.Inputs.create_inducing = project, lib/cov_se_fat.ml:220
        z0 = params.tproj.T @ z0
    z0 = np.asfortranarray(z0)
    k1, z1, s2, le, nev = optim.train(F, cov_se_fat, kernel, z0, x, y, sigma2=sigma2,
                                      max_iter=60 if max_iter is None else max_iter)
    if verbose:
        print("log evidence %.6f after %d evaluations, sigma2 %.6g" % (le, nev, s2))
    E = F.Eval
    inducing = E.Inducing.calc(k1, z1)
    model = E.Model.calc(E.Inputs.calc(x, inducing), sigma2=s2)
    trained = E.Trained.calc(model, targets=y)
    out = SimpleNamespace(
        sigma2=float(s2), target_mean=target_mean, input_means=input_means, input_stddevs=input_stddevs,
        kernel=k1, inducing_points=np.asfortranarray(z1), coeffs=E.Trained.calc_mean_coeffs(trained).copy(),
        co_variance_coeffs=E.Model.calc_co_variance_coeffs(model), log_evidence=float(le),
        stats=E.Stats.calc(trained))
    if functor is None:
        GP.close()
    return out


def save_model(path_or_file, model):
    """write_model, bin/ocaml_gpr.ml:203-232."""
    p = model.kernel.params
    fields = dict(format=FORMAT, cov="se_fat", sigma2=model.sigma2, target_mean=model.target_mean,
                  input_means=model.input_means, input_stddevs=model.input_stddevs, d=p.d, log_sf2=p.log_sf2,
                  inducing_points=model.inducing_points, coeffs=model.coeffs,
                  chol_km=np.triu(model.co_variance_coeffs[0]), r_mat=np.triu(model.co_variance_coeffs[1]))
    if p.tproj is not None:
        fields["tproj"] = p.tproj
    if p.log_hetero_skedasticity is not None:
        fields["log_hetero_skedasticity"] = p.log_hetero_skedasticity
    if p.log_multiscales_m05 is not None:
        fields["log_multiscales_m05"] = p.log_multiscales_m05
    np.savez(path_or_file, **fields)


def load_model(path_or_file):
    """read_model, bin/ocaml_gpr.ml:367-371."""
    z = np.load(path_or_file, allow_pickle=False)
    if str(z["format"]) != FORMAT or str(z["cov"]) != "se_fat":
        raise ValueError("not a %s model file" % FORMAT)
    params = cov_se_fat.Params.create(
        int(z["d"]), float(z["log_sf2"]), z["tproj"] if "tproj" in z else None,
        z["log_hetero_skedasticity"] if "log_hetero_skedasticity" in z else None,
        z["log_multiscales_m05"] if "log_multiscales_m05" in z else None)
    return SimpleNamespace(
        sigma2=float(z["sigma2"]), target_mean=float(z["target_mean"]), input_means=z["input_means"],
        input_stddevs=z["input_stddevs"], kernel=cov_se_fat.Kernel.create(params),
        inducing_points=np.asfortranarray(z["inducing_points"]), coeffs=z["coeffs"],
        co_variance_coeffs=(np.asfortranarray(z["chol_km"]), np.asfortranarray(z["r_mat"])))


def predict(model, inputs, with_stddev=False, predictive=True, functor=None):
    """The `test` command (bin/ocaml_gpr.ml:373-413): standardise the inputs with the stored statistics, predict
    means from (inducing_points, coeffs) and, with_stddev, standard deviations from the stored co-variance
    coefficients; means are shifted back by target_mean.  Returns means, or (means, stddevs)."""
    x = np.asarray(inputs, dtype=np.float64)
    big_dim = model.input_stddevs.shape[0]
    if x.shape[0] != big_dim:
        raise ValueError("incompatible dimension of inputs (%d), expected %d" % (x.shape[0], big_dim))
    x = apply_standardization(x, model.input_means, model.input_stddevs)
    GP = functor if functor is not None else fitc_gp.Make_deriv(cov_se_fat)
    E = GP.Variational_FIC.Eval
    mean_predictor = E.Mean_predictor.calc(model.inducing_points, model.coeffs)
    inducing = E.Inducing.calc(model.kernel, model.inducing_points)
    tin = E.Inputs.calc(x, inducing)
    means = E.Means.get(E.Means.calc(mean_predictor, tin)) + model.target_mean
    out = means
    if with_stddev:
        cvp = E.Co_variance_predictor.calc(model.kernel, model.inducing_points, model.co_variance_coeffs)
        v = E.Variances.get(E.Variances.calc(cvp, model.sigma2, tin), predictive=predictive)
        out = (means, np.sqrt(v))
    if functor is None:
        GP.close()
    return out


def format_predictions(means, stddevs=None):
    """The tool's output lines (bin/ocaml_gpr.ml:403-413): "%f" or "%f,%f"."""
    buf = io.StringIO()
    if stddevs is None:
        for mu in means:
            buf.write("%f\n" % mu)
    else:
        for mu, sd in zip(means, stddevs):
            buf.write("%f,%f\n" % (mu, sd))
    return buf.getvalue()
