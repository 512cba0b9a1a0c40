#!/bin/bash
# FETCH_SIZE of the two SYRK launches at C2 over repeated profiler passes, slice groups of 16 (default) against slice by
# slice (GPRHIP_SYRK_GROUP=1): the figure moves from pass to pass with the drift of the workgroups of a round.
root=$(pwd); out=$root/gpurun_out/lab12; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for rep in 1 2 3; do for g in 16 1; do
  export GPRHIP_SYRK_GROUP=$g
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/f_${g}_$rep -- python3 $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-configs > /dev/null 2>&1
  db=$(ls $out/f_${g}_$rep/*/*.db | head -1)
  python3 - "$db" "$g" <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select kernel_name, value from counters_collection where counter_name='FETCH_SIZE' and kernel_name like '%gemm_f64_tn_w%'").fetchall()
agg = {}
for k, v in rows:
    agg.setdefault(k.split('(')[0], []).append(2.0 * v * 1024 / 1e9)
print("group", sys.argv[2], {k: [round(x, 1) for x in v] for k, v in agg.items()})
PY
  rm -rf $out/f_${g}_$rep
done; done
