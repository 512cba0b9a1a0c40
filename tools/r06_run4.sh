#!/bin/bash
set -u
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden or mid or thresholds or degenerate or two_shards or rccl_with_one_rank or fp32_bulk_refuses or chunking or update_sigma2" 2>&1 | tail -15 > gpurun_out/r06d_tests.log
tail -15 gpurun_out/r06d_tests.log
python3 tools/latency.py 2000,128,3 1280,128,4 2560,256,8 10000,256,8 100000,128,8 > gpurun_out/r06d_latency.txt 2>&1
cat gpurun_out/r06d_latency.txt
bash tools/trace_midsize.sh 2000 128 3 > gpurun_out/r06d_timeline_n2000_m128.txt 2>&1
bash tools/trace_midsize.sh 2560 256 8 > gpurun_out/r06d_timeline_n2560_m256.txt 2>&1
cat gpurun_out/r06d_timeline_n2000_m128.txt gpurun_out/r06d_timeline_n2560_m256.txt
