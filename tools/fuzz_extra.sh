#!/bin/bash
# Extra-long random-shape parity sweeps beyond tools/fuzz_round.sh (seed ranges nobody has run before); log -> gpurun_out/<tag>_fuzz_extra.txt
#   usage (GPU box, repo root): bash tools/fuzz_extra.sh <tag>
tag=${1:-rXX}
out=gpurun_out/${tag}_fuzz_extra.txt
: > $out
run() { echo "== $*" >> $out; env "$@" python -m pytest tests/test_gpu_parity.py -q -s -m gpu -k "long_sweep" 2>&1 | grep -E "sweep:|passed|failed" >> $out; }
run GPR_FUZZ_SEEDS=5000:8000 GPR_FUZZ_SMALL=1
run GPR_FUZZ_SEEDS=8000:8400 GPR_FUZZ_SMALL=1 GPR_FUZZ_SHARDS=5
run GPR_FUZZ_SEEDS=1000:1800
cat $out
