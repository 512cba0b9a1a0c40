#!/bin/bash
for lab in 0 1 2 4 8 16 7 15 31 64; do
  echo "lab=$lab: $(GPRHIP_MID_LAB=$lab REPS=10 python3 tools/latency.py 2000,128,3 2>&1 | sed 's/.*p2_mid.: \([0-9.]*\).*/p2_mid \1/')"
done
