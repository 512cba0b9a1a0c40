#!/bin/bash
# The long random-shape parity sweeps (tests/test_gpu_parity.py, skipped in the default suite); logs go to profiles/.
#   usage (GPU box, repo root): bash tools/fuzz_round.sh <tag>
tag=${1:-rXX}
out=gpurun_out/${tag}_fuzz.txt
: > $out
run() { echo "== $*" >> $out; env "$@" python -m pytest tests/test_gpu_parity.py -q -s -m gpu -k "long_sweep" 2>&1 | grep -E "sweep:|passed|failed" >> $out; }
run GPR_FUZZ_SEEDS=24:224
run GPR_FUZZ_SEEDS=224:324 GPR_FUZZ_SHARDS=6
run GPR_FUZZ_SEEDS=4000:4300 GPR_FUZZ_SMALL=1
run GPR_FUZZ_SEEDS=4300:4400 GPR_FUZZ_SMALL=1 GPR_FUZZ_SHARDS=4
run GPR_FUZZ_F32=8:108
run GPR_FUZZ_POSTERIOR=8:108
cat $out
