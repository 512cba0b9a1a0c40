#!/bin/bash
set -u
root=$(pwd)
run() {
python3 $root/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-configs 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$1', d['ms_per_step'], s['p1_trmm_V'], s['p2_trmm_Q'], s['p2_trmm_S'], s['p2_trmm_X'], s['p1_cov'], s['p2_grad'], d['last_eval'])"
}
for c in 131072 250112 500096 1000064; do GPRHIP_CHUNK_ROWS=$c run chunk$c; done
