#!/bin/bash
# Collects the evidence behind bench.py's roofline block on the GPU box (run through gpurun from the repo root):
#   1. the bench line itself
#   2. rocprofv3 --kernel-trace --stats of the same command (per-kernel average durations)
#   3. three separate --pmc passes (FETCH_SIZE, WRITE_SIZE, MFMA busy), as MI355X_MICROARCH.md prescribes
# Outputs land in gpurun_out/<tag>_*; copy the summaries you want judged into profiles/.
#   usage: bash tools/profile_round.sh <tag>
set -u
tag=${1:-rXX}
root=$(pwd)
out=$root/gpurun_out
mkdir -p "$out"
python3 bench.py --steps 5 --warmup 2 > "$out/${tag}_bench.json" 2> "$out/${tag}_bench.err"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$out/${tag}_stats" -- python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-configs > "$out/${tag}_stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$out/${tag}_pmc_f" -- python3 "$root/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-configs > "$out/${tag}_pmc_f.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$out/${tag}_pmc_w" -- python3 "$root/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-configs > "$out/${tag}_pmc_w.log" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_F64 --kernel-trace -d "$out/${tag}_pmc_m" -- python3 "$root/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-configs > "$out/${tag}_pmc_m.log" 2>&1
cd "$root"
db() { ls "$out/$1"/*/*.db 2>/dev/null | head -1 || ls "$out/$1"/*.db | head -1; }
python3 tools/rocpd_summary.py "$(db ${tag}_stats)" "$out/${tag}_kernel_stats.txt" > /dev/null
python3 tools/pmc_summary.py "$(db ${tag}_pmc_f)" "$(db ${tag}_pmc_w)" "$(db ${tag}_pmc_m)" "$out/${tag}_pmc.json" > /dev/null
# keep the merged-back payload small: the databases are not needed once summarised
rm -rf "$out/${tag}_stats" "$out/${tag}_pmc_f" "$out/${tag}_pmc_w" "$out/${tag}_pmc_m"
tail -c 600 "$out/${tag}_bench.json"; echo; head -8 "$out/${tag}_kernel_stats.txt"
