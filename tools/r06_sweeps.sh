#!/bin/bash
# Random-shape sweeps with every achieved error, its scales and the case's condition estimate logged (tests/margins.py)
set -u
tag=${1:-r06_sweep}
export GPR_MARGINS_LOG=$PWD/gpurun_out/${tag}_margins.jsonl
mkdir -p gpurun_out; : > $GPR_MARGINS_LOG
out=gpurun_out/${tag}.txt; : > $out
run() { echo "== $*" >> $out; env "$@" python -m pytest tests/test_gpu_parity.py -q -s -m gpu -k "long_sweep" 2>&1 | grep -E "sweep:|passed|failed" | cut -c1-600 >> $out; }
run GPR_FUZZ_SEEDS=8000:8200
run GPR_FUZZ_SEEDS=8200:8260 GPR_FUZZ_SHARDS=5
run GPR_FUZZ_SEEDS=9000:9200 GPR_FUZZ_SMALL=1
cat $out
