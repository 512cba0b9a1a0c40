#!/bin/bash
# Kernel timeline (rocprofv3 --kernel-trace) of the last gradient evaluation at a mid-size shape.
#   usage (GPU box, repo root): bash tools/trace_midsize.sh N M D > out.txt
set -u
root=$(pwd); out=$root/gpurun_out/trace_mid; rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
REPS=2 rocprofv3 --kernel-trace --output-format csv -d "$out" -- python3 "$root/tools/one_eval.py" "$1" "$2" "$3" > "$out/run.log" 2>&1
cd "$root"
csv=$(ls "$out"/*/*kernel_trace.csv 2>/dev/null | head -1)
python3 tools/trace_last_eval.py "$csv" cov_upper
rm -rf "$out"
