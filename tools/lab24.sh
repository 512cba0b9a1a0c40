#!/bin/bash
export GPRHIP_LIBRARY=lab  # (the round barrier lives in the lab build since round 6: make -C gpr_amd/csrc lab)
# Round 5: the round barrier of the fp32 SYRK-shaped launches (mfma_gemm.hip, GPRHIP_ROUND_SYNC_US) on BASELINE config 3
# (n = 1M, m = 4096, d = 32, fp32 bulk): evaluation and stage times with the barrier off (0) and at several bounds.
#   usage (GPU box, repo root): bash tools/lab24.sh
export PREC=f32
for us in 0 50 0 20 100 50; do
  echo "GPRHIP_ROUND_SYNC_US=$us"
  GPRHIP_ROUND_SYNC_US=$us python3 tools/run_config3.py 2>&1 | grep -v amdgpu.ids
done
