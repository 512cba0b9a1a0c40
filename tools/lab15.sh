#!/bin/bash
# Round 4: how much of an engine launch is memory latency that the two-buffer LDS ring does not hide?
# GPRHIP_LAB_NOSTEP=1 keeps the operand pointers of every k-loop on their first stage (timing only -- the results are
# wrong): all refills after the first hit the L2.  Same instruction stream, same barriers, same epilogues.
for rep in 1 2; do
  echo "default"; python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-configs 2>/dev/null | python3 tools/stage_times.py gt 50
  echo "NOSTEP"; GPRHIP_LAB=1 GPRHIP_LAB_NOSTEP=1 python3 tools/lab15.py
done
