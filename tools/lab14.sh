#!/bin/bash
# Round 4: (a) stage times of the headline with the NT kernels split by epilogue (no scratch spills);
# (b) the same engine at ONE workgroup per CU (GPRHIP_LDS_PAD forces it: one wavefront per SIMD, nothing covers a
#     wave's waits) -- how much of a stage the waits are when they are fully exposed.
for rep in 1 2; do
  echo "default"; python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-configs 2>/dev/null | python3 tools/stage_times.py gt 50
  echo "LDS_PAD=16384 (1 workgroup per CU)"; GPRHIP_LDS_PAD=16384 python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-configs 2>/dev/null | python3 tools/stage_times.py gt 50
done
