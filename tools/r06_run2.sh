#!/bin/bash
# second measurement pass of round 6: whole GPU suite with margins (condition-aware), sweeps incl. mid shapes, timeline of a mid evaluation
set -u
export GPR_MARGINS_LOG=$PWD/gpurun_out/r06b_margins.jsonl
mkdir -p gpurun_out; : > $GPR_MARGINS_LOG
python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r06b_gputest.log
out=gpurun_out/r06b_fuzz.txt; : > $out
run() { echo "== $*" >> $out; env "$@" python -m pytest tests/test_gpu_parity.py -q -s -m gpu -k "long_sweep" 2>&1 | grep -E "sweep:|passed|failed" | cut -c1-900 >> $out; }
run GPR_FUZZ_SEEDS=8000:8200
run GPR_FUZZ_SEEDS=8200:8260 GPR_FUZZ_SHARDS=5
run GPR_FUZZ_SEEDS=9000:9150 GPR_FUZZ_SMALL=1
run GPR_FUZZ_SEEDS=9500:9800 GPR_FUZZ_MID=1
run GPR_FUZZ_SEEDS=9800:9860 GPR_FUZZ_MID=1 GPR_FUZZ_SHARDS=4
run GPR_FUZZ_F32=400:460
unset GPR_MARGINS_LOG
python3 tools/parity_margins.py gpurun_out/r06b_margins.jsonl > gpurun_out/r06b_parity_margins.txt
bash tools/trace_midsize.sh 2000 128 3 > gpurun_out/r06b_timeline_n2000_m128.txt 2>&1
tail -6 gpurun_out/r06b_gputest.log; cat $out; tail -45 gpurun_out/r06b_parity_margins.txt; cat gpurun_out/r06b_timeline_n2000_m128.txt
