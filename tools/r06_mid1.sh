#!/bin/bash
set -u
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden or mid or thresholds or degenerate or error_behaviour or bench_config or refused" 2>&1 | tail -25 > gpurun_out/r06_mid1_tests.log
tail -25 gpurun_out/r06_mid1_tests.log
python3 tools/latency.py 2000,50,3 2000,128,3 1280,128,4 5000,100,8 100000,128,8 > gpurun_out/r06_mid1_latency.txt 2>&1
cat gpurun_out/r06_mid1_latency.txt
bash tools/r06_sweeps.sh r06_sweep1 > /dev/null 2>&1
tail -3 gpurun_out/r06_sweep1.txt | cut -c1-300
