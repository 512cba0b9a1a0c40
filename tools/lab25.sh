#!/bin/bash
export GPRHIP_LIBRARY=lab  # (the round barrier lives in the lab build since round 6: make -C gpr_amd/csrc lab)
# Round 5: fabric traffic (FETCH_SIZE) of config 3's fp32 launches with the round barrier off / on (tools/lab24.sh times them).
#   usage (GPU box, repo root): bash tools/lab25.sh
set -u
root=$(pwd); out=$root/gpurun_out; mkdir -p "$out"
export PREC=f32
cd /tmp && export TMPDIR=/tmp
for us in 0 50; do
  export GPRHIP_ROUND_SYNC_US=$us
  rm -rf "$out/lab25_f_$us"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$out/lab25_f_$us" -- python3 "$root/tools/run_config3.py" > "$out/lab25_f_$us.log" 2>&1
done
cd "$root"
db() { ls "$out/$1"/*/*.db 2>/dev/null | head -1 || ls "$out/$1"/*.db | head -1; }
for us in 0 50; do
  echo "GPRHIP_ROUND_SYNC_US=$us"
  python3 tools/rocpd_summary.py "$(db lab25_f_$us)" 2>&1 | grep -E "gemm_f32" | cut -c1-200
  rm -rf "$out/lab25_f_$us"
done
