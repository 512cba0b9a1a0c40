import glob
import os

import numpy as np

from oracle import fitc_oracle as O

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def golden_names():
    """Evidence + gradient fixtures."""
    return [n for n in _names() if not n.startswith("posterior_") and not n.startswith("illcond_")]


def illcond_golden_names():
    """Jitter-dominated K_m (ell = e): the regime SURVEY.md 7 singles out; own stated tolerances."""
    return [n for n in _names() if n.startswith("illcond_")]


def posterior_golden_names():
    """Prediction / covariance / sampler / stats fixtures (make_golden.save_posterior)."""
    return [n for n in _names() if n.startswith("posterior_")]


STAT_KEYS = ("n_samples", "target_variance", "sse", "mse", "rmse", "smse", "msll", "mad", "maxad")


def load_golden(name):
    g = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False))
    g["kind"] = str(g["kind"])
    return g


def oracle_kernel(g):
    if g["kind"] == "iso":
        return O.SeIsoKernel(float(g["log_ell"]), float(g["log_sf2"]))
    return O.SeFatKernel(int(g["d"]), float(g["log_sf2"]), g.get("tproj"), g.get("log_hetero"), g.get("log_multiscales"))


def synth(seed, n, m, d):
    """BASELINE.md section 2 synthetic generator."""
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(d, n))
    y = np.sin(X.sum(0)) + 0.1 * rng.normal(size=n)
    Z = X[:, rng.permutation(n)[:m]] + 0.01 * rng.normal(size=(d, m))
    return np.asfortranarray(X), y, np.asfortranarray(Z)


def relinf(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300))
