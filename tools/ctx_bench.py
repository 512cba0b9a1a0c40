"""The single-process multi-device context (gprhip_ctx_* / gprhip_sharded_*) timed at a BASELINE shape.
    usage (GPU box, repo root): python3 tools/ctx_bench.py [--devices 0,1,...] [--points N --inducing M --dims D]
`--devices 0,0,0,0` (one device named several times) is the validation mode: the shards share the device, so the time
is the sum of the shards' work plus the host-side cost of driving them -- compared with the plain single-problem
evaluation of the same shape it bounds the context's own overhead.  With distinct devices the exchange is RCCL.
Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpr_amd  # noqa: E402
from bench import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--devices", default="0")
ap.add_argument("--points", type=int, default=1_000_000)
ap.add_argument("--inducing", type=int, default=2048)
ap.add_argument("--dims", type=int, default=8)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--plain", action="store_true", help="also time gpr_amd.Problem on the first device")
args = ap.parse_args()
devs = [int(x) for x in args.devices.split(",")]
n, m, d = args.points, args.inducing, args.dims
X, y, Z0 = synth(2, n, m, d)
rng = np.random.default_rng(1234)


def hypers():
    return dict(log_ell=0.5 * np.log(d) + 1e-3 * rng.normal(), log_sf2=1e-3 * rng.normal(),
                sigma2=0.1 * np.exp(1e-3 * rng.normal()), inducing=Z0 + 1e-3 * rng.normal(size=Z0.shape))


def timed(obj):
    for _ in range(2):
        obj.eval(**hypers())
    ts = []
    for _ in range(args.steps):
        h = hypers()
        t0 = time.perf_counter()
        ev = obj.eval(**h)
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3, ev


out = {"devices": devs, "n": n, "m": m, "d": d}
ctx = gpr_amd.Context(devs)
sp = gpr_amd.ShardedDeviceProblem(ctx, gpr_amd.COV_SE_ISO, n, d, d, m)
sp.set_inputs(X)
sp.set_targets(y)
sp.set_timing(1)
out["comm_mode"] = {0: "none", 1: "rccl", 2: "same-device sum"}[ctx.comm_mode]
out["ctx_ms_per_eval"], ev = timed(sp)
out["points_per_s"] = n / out["ctx_ms_per_eval"] * 1e3
out["comm"] = sp.comm_stats()
out["shards"] = [sp.shard(i) for i in range(ctx.ndev)]
out["l"] = float(ev.l)
sp.close()
ctx.close()
if args.plain:
    p = gpr_amd.Problem(gpr_amd.COV_SE_ISO, n, d, d, m, device=devs[0])
    p.set_inputs(X)
    p.set_targets(y)
    out["plain_ms_per_eval"], ev1 = timed(p)
    p.close()
print(json.dumps(out))
