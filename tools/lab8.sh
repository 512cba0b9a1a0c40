#!/bin/bash
set -u
root=$(pwd)
./build/gemm_check | grep "syrk\|failed"; F32=1 ./build/gemm_check | grep "syrk\|failed"
run() {
python3 $root/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-configs 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$1', d['ms_per_step'], s['p1_syrk_B'], s['p2_syrk_W'], d['last_eval'])"
}
for r in 0.76 0.80 0.84 0.88; do GPRHIP_SY_RATIO_WS=$r run c2_ws$r; done
for r in "0.76 0.85" "0.70 0.85" "0.82 0.85" "0.76 0.80" "0.76 0.90" "0.76 0.95"; do set -- $r; echo "f32 W=$1 WS=$2"; GPRHIP_SY_RATIO=$1 GPRHIP_SY_RATIO_WS=$2 PREC=f32 python3 tools/run_config3.py | python3 -c "
import sys,ast
l=sys.stdin.read().splitlines(); print(l[0]); d=ast.literal_eval(l[1].strip()); print('   syrk_B',d['p1_syrk_B'],'syrk_W',d['p2_syrk_W'])"; done
