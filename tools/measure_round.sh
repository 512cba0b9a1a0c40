#!/bin/bash
# A round's measurements on one box: the bench line with rocprofv3 kernel statistics and PMC passes (profile_round.sh), the
# whole-evaluation latency table at small and medium sizes, per-GPU shard sizes of the 4 / 8-way split, kernel timelines of the
# one- and two-tile mid-size evaluations.  usage (GPU box, repo root): bash tools/measure_round.sh <tag>
set -u
tag=${1:-rXX}
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden or mid or thresholds or degenerate or error_behaviour or large_common" 2>&1 | tail -3 > gpurun_out/${tag}_quick_tests.log
bash tools/profile_round.sh ${tag} > gpurun_out/${tag}_profile_round.log 2>&1
python3 tools/latency.py 2000,50,3 1000,10,1 100000,50,3 1000000,50,8 2000,128,3 1280,128,4 100000,128,8 2000,200,6 2560,256,8 10000,256,8 5000,200,4 20000,384,8 50000,512,8 60000,768,8 100000,1024,8 16384,2048,8 125000,2048,8 24000,4096,8 > gpurun_out/${tag}_latency.txt 2>&1
bash tools/trace_midsize.sh 2000 128 3 > gpurun_out/${tag}_timeline_n2000_m128.txt 2>&1
bash tools/trace_midsize.sh 2560 256 8 > gpurun_out/${tag}_timeline_n2560_m256.txt 2>&1
bash tools/trace_midsize.sh 10000 256 8 > gpurun_out/${tag}_timeline_n10000_m256.txt 2>&1
cat gpurun_out/${tag}_quick_tests.log gpurun_out/${tag}_latency.txt; tail -c 1500 gpurun_out/${tag}_bench.json; echo; head -12 gpurun_out/${tag}_kernel_stats.txt
