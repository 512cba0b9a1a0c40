"""Parity checks that (i) compare a gradient family by family, each block of the reference's `Hyper.get_all` order against
ITS OWN largest entry (lib/cov_se_iso.ml:188-202: [Log_ell; Log_sf2; inducing (ind-major)]; lib/cov_se_fat.ml:290-342:
[Log_sf2; inducing (ind-major); Proj (big_dim-major); Log_hetero_skedasticity; Log_multiscale_m05]) -- one max-norm over the
whole vector lets the entries of order 1e6 (Log_ell, Log_sf2 at m = 2048) hide percent-level errors in the inducing block --
and (ii) record every achieved error beside its bound: with GPR_MARGINS_LOG=<file> each check appends one JSON line
{test, what, err, tol}; tools/parity_margins.py turns the log of a whole `pytest -m gpu` run into the table
profiles/r06_parity_margins.txt that the tolerances in tests/test_gpu_parity.py are derived from (<= 10 x the worst observed).
"""
import json
import os

import numpy as np


_context = {}


def note(_reset=True, **kw):
    """Facts about the case in hand (condition estimate, shape ...) that every following record carries."""
    if _reset:
        _context.clear()
    _context.update(kw)


def _record(what, err, tol, **extra):
    path = os.environ.get("GPR_MARGINS_LOG")
    if not path:
        return
    test = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]
    rec = {"test": test, "what": what, "err": float(err), "tol": float(tol)}
    rec.update(_context)
    rec.update(extra)
    with open(path, "a") as f:
        f.write(json.dumps(rec) + "\n")


def families(kind, d, m, D=0, proj=False, het=False, ms=False):
    """[(family, slice)] of the gradient vector in the reference's Hyper.get_all order."""
    out, pos = [], 0

    def take(name, count):
        nonlocal pos
        out.append((name, slice(pos, pos + count)))
        pos += count

    if kind == "iso":
        take("log_ell", 1)
        take("log_sf2", 1)
        take("inducing", d * m)
    else:
        take("log_sf2", 1)
        take("inducing", d * m)
        if proj:
            take("proj", D * d)
        if het:
            take("hetero", m)
        if ms:
            take("multiscale", d * m)
    return out


def families_golden(g):
    d, m = g["Z"].shape
    return families(g["kind"], d, m, D=g["X"].shape[0], proj="tproj" in g, het="log_hetero" in g, ms="log_multiscales" in g)


def relinf(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


EPS64 = 2.0 ** -53
EPS32 = 2.0 ** -24
# conditioning allowance = ALLOW_FACTOR x cond(K_m + jitter) x unit roundoff, relative to the gradient's largest entry.  Over
# 4700 fp64 family checks with a condition estimate (profiles/r06_parity_margins.txt) the absolute error of a family reached
# 5.3 x cond x 2^-53 of the largest entry (seed 9698: 108 inducing points on a line, cond 6e7); 8 covers it with margin 1.5.
ALLOW_FACTOR = 8.0


def family_errors(got, ref, fams, detail=None, allow=0.0):
    """{family: max-abs error relative to that family's largest reference entry}.  A family whose reference entries all
    vanish to rounding (below 1e-12 of the vector's largest entry: e.g. d-th coordinates that the kernel does not see) is
    measured against the whole vector's scale instead -- there is no scale of its own to be relative to.
    allow: the conditioning allowance, as a fraction of the vector's largest entry, taken off every family's absolute error
    first -- cond(K_m + jitter) x unit roundoff.  The trace terms of a gradient entry sum entries of W = U^-1 W~ U^-T, which
    are |U^-1|^2 ~ cond times larger than the sum itself (with K_m jitter-dominated -- points on a line, more inducing than
    training points -- the inducing-point entries of the gradient all but vanish and what is left of them is this rounding
    error: measured against 80-bit central differences the oracle's own entries are then off by 1e-4 of their family's
    largest, test/oracle notes in DESIGN section 6); a family-relative bound alone cannot hold there for ANY evaluation
    order, the reference's included."""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert fams[-1][1].stop == ref.shape[0], (fams, ref.shape)
    whole = max(float(np.max(np.abs(ref))), 1e-300)
    out = {}
    for name, sl in fams:
        r = ref[sl]
        if r.size == 0:
            continue
        scale = float(np.max(np.abs(r)))
        if scale < 1e-12 * whole:
            scale = whole
        raw = float(np.max(np.abs(got[sl] - r)))
        out[name] = max(0.0, raw - allow * whole) / scale
        if detail is not None:
            detail[name] = dict(scale=scale, whole=whole, raw=raw / scale, allow=allow)
    return out


def check_grad(got, ref, fams, tol, what="grad", cond=None, unit=EPS64):
    """Every family within tol of its own largest entry; cond (the device's 2-norm estimate of cond(K_m + jitter), where the
    caller has it) adds the conditioning allowance ALLOW_FACTOR x cond x unit relative to the vector's largest entry (see
    family_errors)."""
    detail = {}
    errs = family_errors(got, ref, fams, detail, allow=(ALLOW_FACTOR * cond * unit if cond else 0.0))
    for name, e in errs.items():
        _record("%s.%s" % (what, name), e, tol, **detail[name])
    bad = {k: v for k, v in errs.items() if not v <= tol}
    assert not bad, "%s: families beyond %.1e: %s (all: %s; cond %s)" % (what, tol, bad, errs, cond)
    return errs


def check_rel(what, got, ref, tol, floor=0.0):
    """|got - ref| <= tol * max(|ref|, floor)"""
    err = abs(float(got) - float(ref)) / max(abs(float(ref)), floor, 1e-300)
    _record(what, err, tol)
    assert err <= tol, "%s: %.3e > %.1e (got %r, ref %r)" % (what, err, tol, got, ref)
    return err


def check_vec(what, got, ref, tol, cond=None, unit=EPS64):
    """max-abs error relative to the largest reference entry; cond (as in check_grad) takes the conditioning allowance
    ALLOW_FACTOR x cond x unit of that entry off first -- the forward error of the triangular solves behind the mean
    coefficients is cond(K_m + jitter) x unit roundoff for every evaluation order, the oracle's included
    (test_mean_coefficients_against_an_80_bit_evaluation)."""
    err = relinf(got, ref)
    if cond:
        err = max(0.0, err - ALLOW_FACTOR * cond * unit)
    _record(what, err, tol)
    assert err <= tol, "%s: relinf %.3e > %.1e (cond %s)" % (what, err, tol, cond)
    return err


# the same three as expressions (for `assert a_ok(...) and b_ok(...)` lines): they assert inside and return True
def grad_ok(got, ref, fams, tol, what="grad", cond=None, unit=EPS64):
    check_grad(got, ref, fams, tol, what, cond, unit)
    return True


def rel_ok(what, got, ref, tol, floor=0.0):
    check_rel(what, got, ref, tol, floor)
    return True


def vec_ok(what, got, ref, tol, cond=None, unit=EPS64):
    check_vec(what, got, ref, tol, cond, unit)
    return True
