#!/bin/bash
# Round 4: fabric traffic of the engine launches at m = 4096 fp64 (the BASELINE config-4 shard), one FETCH_SIZE pass.
root=$(pwd); out=$root/gpurun_out/lab23; mkdir -p $out
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/f -- python3 $root/tools/run_config4.py > /dev/null 2>&1)
db=$(ls $out/f/*/*.db | head -1)
python3 - "$db" <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select kernel_name, grid_size_x, value, duration from counters_collection where counter_name='FETCH_SIZE' and kernel_name like '%gemm_f64%'").fetchall()
agg = {}
for k, g, v, d in rows:
    agg.setdefault((k.split('(')[0], g), []).append((2.0 * v * 1024 / 1e9, d / 1e6))
for k, v in agg.items():
    if len(v) >= 3 and sum(x[1] for x in v) / len(v) > 1.0:
        print(k, "n=%d  fetch %.1f GB  %.2f ms" % (len(v), sum(x[0] for x in v) / len(v), sum(x[1] for x in v) / len(v)))
PY
rm -rf $out
