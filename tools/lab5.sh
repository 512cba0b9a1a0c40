#!/bin/bash
set -u
root=$(pwd)
LAB=1 ./build/gemm_check | grep "^time" | tail -9
run() {
python3 $root/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-configs 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$1', d['ms_per_step'], s['p1_syrk_B'], s['p2_syrk_W'], d['last_eval'])"
}
for r in 0.70 0.76 0.82; do GPRHIP_SY_RATIO=$r run ratio$r; done
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -5
