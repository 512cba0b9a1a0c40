#!/bin/bash
set -u
root=$(pwd)
run() {
python3 $root/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-configs 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$1', d['ms_per_step'], 'grad', s['p2_grad'], 'cov', s['p1_cov'], d['last_eval'])"
}
run new
python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or offset or scalar_and_mfma or headline or c4 or wide or dimensions" 2>&1 | tail -3
