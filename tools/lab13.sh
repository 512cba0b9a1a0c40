#!/bin/bash
# SYRK k-slice length at C2: fabric traffic (FETCH_SIZE, one profiler pass each) and launch time.
root=$(pwd); out=$root/gpurun_out/lab13; mkdir -p $out
for r in 2048 4096 8192 16384; do
  export GPRHIP_SLICE_ROWS=$r
  echo "SLICE_ROWS=$r"
  python3 $root/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-configs 2>/dev/null | python3 $root/tools/stage_times.py gt 56 | cut -c1-160
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/f_$r -- python3 $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-configs > /dev/null 2>&1)
  db=$(ls $out/f_$r/*/*.db | head -1)
  python3 - "$db" <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select kernel_name, value from counters_collection where counter_name='FETCH_SIZE' and kernel_name like '%gemm_f64_tn_w%'").fetchall()
agg = {}
for k, v in rows:
    agg.setdefault(k.split('(')[0], []).append(2.0 * v * 1024 / 1e9)
print("   fetch GB", {k: round(sum(v) / len(v), 1) for k, v in agg.items()})
PY
  rm -rf $out/f_$r
done
