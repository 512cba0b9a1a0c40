"""Print ms/step and the per-stage times of a bench.py JSON line read from stdin (optionally only stages
below/above a threshold in ms: `python tools/stage_times.py lt 3`)."""
import json
import sys

j = json.loads(sys.stdin.read().strip().splitlines()[-1])
op, thr = (sys.argv[1], float(sys.argv[2])) if len(sys.argv) > 2 else ("gt", -1.0)
st = {k: round(v, 2) for k, v in j.get("stage_ms", {}).items() if (v < thr if op == "lt" else v > thr)}
print(round(j["value"]), round(j["ms_per_step"], 2), st)
