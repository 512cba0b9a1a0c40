"""Three gradient evaluations of one shape (for kernel timelines).  usage: python3 tools/one_eval.py [n m d]  (default: N from the
environment or 16384, m = 2048, d = 8)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpr_amd  # noqa: E402
from bench import synth  # noqa: E402

n, m, d = int(os.environ.get("N", 16384)), 2048, 8
if len(sys.argv) >= 4:
    n, m, d = (int(v) for v in sys.argv[1:4])
X, y, Z = synth(2, n, m, d)
p = gpr_amd.Problem(gpr_amd.COV_SE_ISO, n, d, d, m)
p.set_inputs(X)
p.set_targets(y)
for _ in range(3):
    ev = p.eval(log_ell=0.5 * np.log(d), log_sf2=0.0, sigma2=0.1, inducing=Z)
print(ev.l)
p.close()
