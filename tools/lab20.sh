#!/bin/bash
# Round 4: the pass-2 SYRK through the kernel of the pass-1 one (gemm_f64_tn_ws: column sums on the diagonal tiles)
for rep in 1 2; do for v in 0 1; do
  export GPRHIP_W_AS_WS=$v
  echo "W_AS_WS=$v"
  python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs 2>/dev/null | python3 tools/stage_times.py gt 56 | cut -c1-160
done; done
