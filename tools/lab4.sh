#!/bin/bash
# SYRK diagonal-tile variant: correctness (fp64 + fp32 harness), 1M-row scaled SYRK with and without it, bench A/B
set -u
root=$(pwd)
./build/gemm_check | grep -v "^time"
F32=1 ./build/gemm_check | grep -v "^time"
for v in 0 1; do echo "GPRHIP_NO_SY=$v"; GPRHIP_NO_SY=$v LAB=1 ./build/gemm_check | grep "syrk"; done
run() {
python3 $root/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-configs 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$1', d['ms_per_step'], s['p1_syrk_B'], s['p2_syrk_W'], d['last_eval'])"
}
GPRHIP_NO_SY=1 run no_sy
for r in 0.60 0.64 0.68 0.72; do GPRHIP_SY_RATIO=$r run ratio$r; done
GPRHIP_NO_SY=1 run no_sy
