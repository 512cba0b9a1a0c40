"""Ad-hoc GPU parity + timing probe (development aid; the real tests live in tests/)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gpr_amd
from oracle import fitc_oracle as O


def synth(seed, n, m, d):
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(d, n))
    y = np.sin(X.sum(0)) + 0.1 * rng.normal(size=n)
    Z = X[:, rng.permutation(n)[:m]] + 0.01 * rng.normal(size=(d, m))
    return np.asfortranarray(X), y, np.asfortranarray(Z)


def compare(seed, n, m, d, chunk=0, variational=False, model_only=False, sigma2=0.1, log_ell=None, precision=0):
    X, y, Z = synth(seed, n, m, d)
    log_ell = 0.5 * np.log(d) if log_ell is None else log_ell
    k = O.SeIsoKernel(log_ell, 0.0)
    ref = O.evaluate_fast(k, Z, X, (0 * y if model_only else y), sigma2, variational=variational)
    p = gpr_amd.Problem(gpr_amd.COV_SE_ISO, n, d, d, m, chunk_rows=chunk, precision=precision)
    p.set_inputs(X); p.set_targets(y)
    ev = p.eval(log_ell=log_ell, log_sf2=0.0, sigma2=sigma2, inducing=Z, variational=variational,
                model_only=model_only)
    ev0 = p.eval(log_ell=log_ell, log_sf2=0.0, sigma2=sigma2, inducing=Z, variational=variational,
                 model_only=model_only, want_grad=False)
    rel = lambda a, b: abs(a - b) / max(abs(b), 1e-300)
    l_ref = ref['l1'] if model_only else ref['l']
    g_err = np.max(np.abs(ev.grad - ref['grad'])) / np.max(np.abs(ref['grad']))
    print("prec=%d " % precision, end="")
    print("n=%d m=%d d=%d chunk=%d var=%d mo=%d: l rel %.2e (nograd %.2e) l1 rel %.2e dls2 rel %.2e grad relinf %.2e coeffs relinf %.2e" % (
        n, m, d, chunk, variational, model_only, rel(ev.l, l_ref), rel(ev0.l, l_ref), rel(ev.l1, ref['l1']),
        rel(ev.dl_dsigma2, ref['dl_dsigma2']), g_err,
        0 if model_only else np.max(np.abs(ev.coeffs - ref['coeffs'])) / np.max(np.abs(ref['coeffs']))))
    p.close()


if __name__ == "__main__":
    compare(1, 300, 5, 3)
    compare(1, 2000, 50, 3)
    compare(2, 2000, 50, 3, variational=True)
    compare(3, 2000, 50, 3, model_only=True)
    compare(4, 5000, 300, 8, chunk=1024)
    compare(5, 3000, 130, 8, chunk=512, variational=True)
    for prec_case in ((1, 2000, 50, 3), (4, 5000, 300, 8), (6, 20000, 512, 16)):
        compare(*prec_case, precision=1)
    compare(3, 2000, 50, 3, model_only=True, precision=1)
    compare(7, 8000, 256, 8, sigma2=1e-3, precision=1)
    if len(sys.argv) > 1:
        n, m, d = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
        os.environ["GPRHIP_TIMING"] = "1"
        X, y, Z = synth(2, n, m, d)
        p = gpr_amd.Problem(gpr_amd.COV_SE_ISO, n, d, d, m, precision=int(os.environ.get('PREC', '0')))
        p.set_inputs(X); p.set_targets(y)
        for it in range(3):
            t0 = time.time()
            ev = p.eval(log_ell=0.5 * np.log(d), log_sf2=0.0, sigma2=0.1, inducing=Z)
            dt = time.time() - t0
            print("eval %d: %.3f s  -> %.3f Mpts/s  l=%.6f" % (it, dt, n / dt / 1e6, ev.l))
        tm = p.last_timings()
        tot = sum(tm.values())
        for k_, v_ in sorted(tm.items(), key=lambda kv: -kv[1]):
            print("   %-12s %9.3f ms  %5.1f%%" % (k_, v_, 100 * v_ / tot))
        F = n * (6.0 * m * m + 4.0 * m * d) + 2.0 * m ** 3
        print("algorithmic TFLOP/s: %.2f (of 78.6)" % (F / dt * 1e-12))
