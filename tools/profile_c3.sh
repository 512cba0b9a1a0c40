#!/bin/bash
# Per-kernel durations (--kernel-trace --stats) and FETCH_SIZE / WRITE_SIZE / MFMA-busy passes of BASELINE config 3 in the fp32-bulk mode (tools/run_config3.py),
# summarised per kernel into gpurun_out/<tag>_pmc_c3.json.   usage (GPU box, repo root): bash tools/profile_c3.sh <tag>
set -u
tag=${1:-rXX}
root=$(pwd); out=$root/gpurun_out; mkdir -p "$out"
export PREC=f32
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$out/${tag}_c3_s" -- python3 "$root/tools/run_config3.py" > "$out/${tag}_c3_s.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$out/${tag}_c3_f" -- python3 "$root/tools/run_config3.py" > "$out/${tag}_c3_f.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$out/${tag}_c3_w" -- python3 "$root/tools/run_config3.py" > "$out/${tag}_c3_w.log" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_F64 --kernel-trace -d "$out/${tag}_c3_m" -- python3 "$root/tools/run_config3.py" > "$out/${tag}_c3_m.log" 2>&1
cd "$root"
db() { ls "$out/$1"/*/*.db 2>/dev/null | head -1 || ls "$out/$1"/*.db | head -1; }
python3 tools/rocpd_summary.py "$(db ${tag}_c3_s)" "$out/${tag}_c3_kernel_stats.txt" > /dev/null
python3 tools/pmc_summary.py "$(db ${tag}_c3_f)" "$(db ${tag}_c3_w)" "$(db ${tag}_c3_m)" "$out/${tag}_pmc_c3.json"
rm -rf "$out/${tag}_c3_s" "$out/${tag}_c3_f" "$out/${tag}_c3_w" "$out/${tag}_c3_m"
