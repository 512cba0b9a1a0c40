#!/bin/bash
set -u
root=$(pwd)
for rep in 1 2; do
python3 $root/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-configs 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('ipw2', d['ms_per_step'], s['p1_trmm_V'], s['p2_trmm_X'])"
GPRHIP_LAB_IPW4=1 python3 $root/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-configs 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('ipw4', d['ms_per_step'], s['p1_trmm_V'], s['p2_trmm_X'])"
done
