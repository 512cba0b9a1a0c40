#!/bin/bash
# Random-shape parity sweeps on the final build of a round (seed ranges disjoint from tools/fuzz_round.sh and
# tools/parity_round.sh); log -> gpurun_out/<tag>_fuzz_final.txt
tag=${1:-rXX}
out=gpurun_out/${tag}_fuzz_final.txt
: > $out
run() { echo "== $*" >> $out; env "$@" python -m pytest tests/test_gpu_parity.py -q -s -m gpu -k "long_sweep" 2>&1 | grep -E "sweep:|passed|failed" | cut -c1-900 >> $out; }
run GPR_FUZZ_SEEDS=6000:6300
run GPR_FUZZ_SEEDS=6300:6400 GPR_FUZZ_SHARDS=5
run GPR_FUZZ_SEEDS=7000:7400 GPR_FUZZ_SMALL=1
run GPR_FUZZ_SEEDS=21000:21600 GPR_FUZZ_MID=1
run GPR_FUZZ_SEEDS=21600:21700 GPR_FUZZ_MID=1 GPR_FUZZ_SHARDS=3
run GPR_FUZZ_F32=200:300
run GPR_FUZZ_POSTERIOR=200:300
cat $out
