#!/bin/bash
# Round 4: pass-2 SYRK through the column-sum kernel (GPRHIP_W_AS_WS=1, default) against the plain weighted kernel (=0)
# at the headline, BASELINE config 3 (fp32 bulk and fp64, m = 4096) and the config-4 shard.
for v in 1 0; do
  export GPRHIP_W_AS_WS=$v
  echo "W_AS_WS=$v  C2"; python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs 2>/dev/null | python3 tools/stage_times.py gt 56 | cut -c1-160
  echo "W_AS_WS=$v  C3"; python3 tools/run_config3.py 2>&1 | cut -c1-220
  echo "W_AS_WS=$v  C4 shard"; python3 tools/run_config4.py 2>&1 | cut -c1-220
done
