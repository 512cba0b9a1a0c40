#!/bin/bash
# Round 4: fabric traffic and time of the two SYRK launches per launch, with the pass-2 one through the plain weighted
# kernel (GPRHIP_W_AS_WS=0: gemm_f64_tn_w, 172 GB) and through the column-sum kernel (=1, default: gemm_f64_tn_ws, 67 GB).
# (The run logged in profiles/r04_lab_syrk_w_traffic.txt compared the plain kernel as is and with the device drained by a
# host synchronisation right before the launch -- no difference.)
root=$(pwd); out=$root/gpurun_out/lab19; mkdir -p $out
for v in 0 1; do
  export GPRHIP_W_AS_WS=$v
  echo "W_AS_WS=$v"
  python3 $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs 2>/dev/null | python3 $root/tools/stage_times.py gt 56 | cut -c1-160
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/f_$v -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs > /dev/null 2>&1)
  db=$(ls $out/f_$v/*/*.db | head -1)
  python3 - "$db" <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select kernel_name, value, duration from counters_collection where counter_name='FETCH_SIZE' and kernel_name like '%gemm_f64_tn_w%' order by dispatch_id").fetchall()
for k, v, d in rows:
    if d > 5e6: print("   ", k.split('(')[0], "fetch %.1f GB" % (2.0 * v * 1024 / 1e9), "%.2f ms" % (d / 1e6))
PY
  rm -rf $out/f_$v
done
