#!/bin/bash
# A/B of the SYRK item order on one box: slices one by one (GPRHIP_SYRK_GROUP=1) against groups of 16 whose full super
# tiles run first (the default).  Measured round 3: pass-1 SYRK 59.25-59.55 -> 58.9-59.1 ms, pass-2 58.8-59.3 -> 58.5-58.7 ms
# at C2 on a slow box; no difference at C3 (fp32, m=4096).
for rep in 1 2 3; do for g in 1 16; do echo "G=$g"; GPRHIP_SYRK_GROUP=$g python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-configs 2>/dev/null | python3 tools/stage_times.py gt 50 | cut -c1-60,60-200 | sed 's/p1_trmm.*p2_syrk_W/p2_syrk_W/' | cut -c1-120; done; done
for g in 1 16; do echo "C3 G=$g"; GPRHIP_SYRK_GROUP=$g PREC=f32 python3 tools/run_config3.py | tail -1 | cut -c1-120; done
