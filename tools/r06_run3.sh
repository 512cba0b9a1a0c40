#!/bin/bash
set -u
export GPR_MARGINS_LOG=$PWD/gpurun_out/r06c_margins.jsonl
mkdir -p gpurun_out; : > $GPR_MARGINS_LOG
python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r06c_gputest.log
unset GPR_MARGINS_LOG
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06c_smoke.log 2>&1
python3 tools/latency.py 2000,128,3 1280,128,4 2000,200,6 2560,256,8 10000,256,8 5000,200,4 > gpurun_out/r06c_latency.txt 2>&1
PREC=f32 python3 tools/run_config3.py > gpurun_out/r06c_c3.txt 2>&1
tail -8 gpurun_out/r06c_gputest.log; cat gpurun_out/r06c_smoke.log gpurun_out/r06c_latency.txt gpurun_out/r06c_c3.txt
