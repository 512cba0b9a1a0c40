#!/bin/bash
# Round 5: mid-size shapes (the reference's default regime, m = min(n/10, 1000)): split-K factor from the launch-time model
# (pick_kslices) and single column tiles instead of pairs for short triangular launches (GPRHIP_PAIR_MIN_ROUNDS; 0 = always
# pairs, as in rounds 2-4).   usage (GPU box, repo root): bash tools/lab26.sh
for r in 0 4 8 16; do
  echo "GPRHIP_PAIR_MIN_ROUNDS=$r"
  GPRHIP_PAIR_MIN_ROUNDS=$r python3 tools/latency.py 10000,256,8 50000,512,8 100000,1024,8 30000,1000,8 16384,2048,8 125000,2048,8 2>&1 | grep -v amdgpu.ids
done
