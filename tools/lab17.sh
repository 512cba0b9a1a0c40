#!/bin/bash
# Round 4: second items of the NN column-tile pairs walked downwards (GPRHIP_NN_DESC=1, default) against upwards (=0):
# stage times of the headline and the fabric traffic (FETCH_SIZE, one profiler pass each) of the V / Q' products.
root=$(pwd); out=$root/gpurun_out/lab17; mkdir -p $out
for rep in 1 2; do for v in 1 0; do
  export GPRHIP_NN_DESC=$v
  echo "NN_DESC=$v"
  python3 $root/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-configs 2>/dev/null | python3 $root/tools/stage_times.py gt 50
done; done
for v in 1 0; do
  export GPRHIP_NN_DESC=$v
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/f_$v -- python3 $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-configs > /dev/null 2>&1)
  db=$(ls $out/f_$v/*/*.db | head -1)
  python3 - "$db" "$v" <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select kernel_name, grid_size_x, value from counters_collection where counter_name='FETCH_SIZE' and (kernel_name like '%gemm_f64_nn%' or kernel_name like '%gemm_f64_nt_sx%')").fetchall()
agg = {}
for k, g, v in rows:
    agg.setdefault((k.split('(')[0], g), []).append(2.0 * v * 1024 / 1e9)
print("NN_DESC", sys.argv[2], {k: (len(v), round(sum(v) / len(v), 2)) for k, v in agg.items() if len(v) > 4})
PY
  rm -rf $out/f_$v
done
