#!/bin/bash
# Round 4: the pass-2 SYRK (gemm_f64_tn_w) sits 0.8 ms above the pass-1 one and fetches 172 GB instead of 67 GB in the
# round-4 profiles: is it the slice-length ratio of its diagonal tiles (GPRHIP_SY_RATIO, tuned 0.76 in round 2)?
for r in 0.70 0.73 0.76 0.79 0.82 0.85; do
  echo "SY_RATIO=$r"
  GPRHIP_SY_RATIO=$r python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs 2>/dev/null | python3 tools/stage_times.py gt 56 | cut -c1-200
done
echo "NO_SY=1"
GPRHIP_NO_SY=1 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs 2>/dev/null | python3 tools/stage_times.py gt 56 | cut -c1-200
