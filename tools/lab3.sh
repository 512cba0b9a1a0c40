#!/bin/bash
# lab: step time vs split-K slice length (1M rows and a 125k-row shard)
set -u
root=$(pwd)
for n in 1000000 125000; do
for sr in 4096 8192 16384 32768; do
  GPRHIP_SLICE_ROWS=$sr python3 $root/bench.py --points $n --steps 6 --warmup 2 --no-cpu-baseline --no-configs 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); s=d.get('stage_ms'); print($n, $sr, round(d['ms_per_step'],2), round(s['p1_syrk_B'],2), round(s['p2_syrk_W'],2), round(d['evidence_only']['ms_per_step'],2))"
done; done
