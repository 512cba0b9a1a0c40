#!/bin/bash
# Round 6, first GPU call: the parity suite and the long sweeps with every achieved error logged beside its bound
# (tests/margins.py -> gpurun_out/r06_margins.jsonl), the mid-size latency table before the round's kernels, config 3 profiles.
set -u
export GPR_MARGINS_LOG=$PWD/gpurun_out/r06_margins.jsonl
mkdir -p gpurun_out; : > $GPR_MARGINS_LOG
python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r06_gputest_first.log
out=gpurun_out/r06_fuzz_first.txt; : > $out
run() { echo "== $*" >> $out; env "$@" python -m pytest tests/test_gpu_parity.py -q -s -m gpu -k "long_sweep" 2>&1 | grep -E "sweep:|passed|failed" >> $out; }
run GPR_FUZZ_SEEDS=8000:8200
run GPR_FUZZ_SEEDS=8200:8260 GPR_FUZZ_SHARDS=5
run GPR_FUZZ_SEEDS=9000:9200 GPR_FUZZ_SMALL=1
run GPR_FUZZ_F32=400:460
run GPR_FUZZ_POSTERIOR=400:440
unset GPR_MARGINS_LOG
python3 tools/parity_margins.py gpurun_out/r06_margins.jsonl > gpurun_out/r06_parity_margins_first.txt
python3 tools/latency.py 2000,50,3 2000,128,3 5000,200,4 10000,256,8 20000,384,8 50000,512,8 60000,768,8 100000,1024,8 125000,2048,8 > gpurun_out/r06_latency_before.txt 2>&1
bash tools/profile_c3.sh r06 > gpurun_out/r06_profile_c3.log 2>&1
tail -5 gpurun_out/r06_gputest_first.log; cat $out; tail -40 gpurun_out/r06_parity_margins_first.txt; cat gpurun_out/r06_latency_before.txt
