#!/bin/bash
# Round 6 parity evidence: the whole GPU suite and the long random-shape sweeps with every achieved error logged beside its
# bound (tests/margins.py) -> gpurun_out/<tag>_gputest.log, <tag>_fuzz.txt, <tag>_parity_margins.txt
set -u
tag=${1:-r06}
export GPR_MARGINS_LOG=$PWD/gpurun_out/${tag}_margins.jsonl
mkdir -p gpurun_out; : > $GPR_MARGINS_LOG
python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/${tag}_gputest.log
out=gpurun_out/${tag}_fuzz.txt; : > $out
run() { echo "== $*" >> $out; env "$@" python -m pytest tests/test_gpu_parity.py -q -s -m gpu -k "long_sweep" 2>&1 | grep -E "sweep:|passed|failed" | cut -c1-900 >> $out; }
run GPR_FUZZ_SEEDS=8000:8200
run GPR_FUZZ_SEEDS=8200:8260 GPR_FUZZ_SHARDS=5
run GPR_FUZZ_SEEDS=9000:9150 GPR_FUZZ_SMALL=1
run GPR_FUZZ_SEEDS=9500:9900 GPR_FUZZ_MID=1
run GPR_FUZZ_SEEDS=9900:9960 GPR_FUZZ_MID=1 GPR_FUZZ_SHARDS=4
run GPR_FUZZ_F32=400:460
run GPR_FUZZ_POSTERIOR=400:440
unset GPR_MARGINS_LOG
python3 tools/parity_margins.py gpurun_out/${tag}_margins.jsonl > gpurun_out/${tag}_parity_margins.txt
tail -6 gpurun_out/${tag}_gputest.log; cat $out
