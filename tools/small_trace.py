"""A few gradient evaluations at the reference's own shape (n=2000, m=50, d=3) for a kernel timeline:
    rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/small -o small --output-format csv -- python3 tools/small_trace.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpr_amd  # noqa: E402
from bench import synth  # noqa: E402

n, m, d = (int(v) for v in os.environ.get("SHAPE", "2000,50,3").split(","))
X, y, Z = synth(1, n, m, d)
p = gpr_amd.Problem(gpr_amd.COV_SE_ISO, n, d, d, m)
p.set_inputs(X)
p.set_targets(y)
for _ in range(int(os.environ.get("EVALS", 6))):
    ev = p.eval(log_ell=0.5 * np.log(d), log_sf2=0.0, sigma2=0.1, inducing=Z)
print(ev.l)
p.close()
