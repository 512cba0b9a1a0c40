"""Fixtures for the 65 .. 256 inducing-point regime (gpr_amd/csrc/mid.hip), generated like tests/golden/make_golden.py from
the CPU oracle's reference-sequence evaluation -- kept in a script of their own so that the older fixtures stay byte for
byte what they were.  The reference's default takes m = min (n / 10) 1000 inducing points (lib/fitc_gp.ml:1474-1479):
n = 1280 gives the m = 128 of `iso_mid`.

    python tests/golden/make_golden_mid.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import fitc_oracle as O  # noqa: E402
from tests.golden.make_golden import save, synth  # noqa: E402


def main():
    X, y, Z = synth(61, 1280, 128, 4)
    le = 0.5 * np.log(4) + 0.1
    save("iso_mid", O.SeIsoKernel(le, 0.2), X, y, Z, 0.08, False, dict(kind="iso", log_ell=le, log_sf2=0.2))
    X, y, Z = synth(62, 1003, 97, 5)
    le = 0.5 * np.log(5)
    save("iso_mid_var", O.SeIsoKernel(le, -0.1), X, y, Z, 0.2, True, dict(kind="iso", log_ell=le, log_sf2=-0.1))
    rng = np.random.default_rng(63)
    n, m, D, d = 900, 90, 7, 4
    Xb = np.asfortranarray(rng.normal(size=(D, n)))
    yb = np.sin(Xb.sum(0)) + 0.1 * rng.normal(size=n)
    P = np.asfortranarray(rng.normal(size=(D, d)) / np.sqrt(D))
    lh = rng.normal(size=m) - 3.0
    kf = O.SeFatKernel(d, 0.1, P, lh)
    Zf = np.asfortranarray(O.se_fat_project(kf, Xb[:, rng.permutation(n)[:m]]) + 0.01 * rng.normal(size=(d, m)))
    save("fat_mid", kf, Xb, yb, Zf, 0.15, True, dict(kind="fat", d=d, log_sf2=0.1, tproj=P, log_hetero=lh))
    # two 128-column tiles (129 .. 256 inducing points): n = 2000 gives the reference's default m = 200
    X, y, Z = synth(64, 2000, 200, 6)
    le = 0.5 * np.log(6) - 0.05
    save("iso_mid2", O.SeIsoKernel(le, 0.1), X, y, Z, 0.1, False, dict(kind="iso", log_ell=le, log_sf2=0.1))
    rng = np.random.default_rng(65)
    n, m, D, d = 1100, 161, 9, 5
    Xb = np.asfortranarray(rng.normal(size=(D, n)))
    yb = np.sin(Xb.sum(0)) + 0.1 * rng.normal(size=n)
    P = np.asfortranarray(rng.normal(size=(D, d)) / np.sqrt(D))
    kf = O.SeFatKernel(d, -0.1, P)
    Zf = np.asfortranarray(O.se_fat_project(kf, Xb[:, rng.permutation(n)[:m]]) + 0.01 * rng.normal(size=(d, m)))
    save("fat_mid2", kf, Xb, yb, Zf, 0.12, False, dict(kind="fat", d=d, log_sf2=-0.1, tproj=P))


if __name__ == "__main__":
    main()
