#!/bin/bash
# Round 4: what does the per-stage workgroup barrier of the engine loop cost?  Rebuilds the library ON THE GPU BOX with
# GEN_NOBARRIER=1 (the generated loops without their s_barrier: timing only, the results are wrong), times the stages of
# the headline shape, and restores the normal build.
set -u
root=$(pwd)
cd gpr_amd/csrc
python3 gen_engine_asm.py engine_asm.inc && make -j8 2>&1 | grep -E "audit_engine: (ok|FAILED)|error"
echo "s_barrier lines in engine_asm.inc: $(grep -c s_barrier engine_asm.inc)"
cd $root
echo "default"; python3 tools/lab15.py
cd gpr_amd/csrc
GEN_NOBARRIER=1 python3 gen_engine_asm.py engine_asm.inc && make -j8 2>&1 | grep -E "audit_engine: (ok|FAILED)|error"
echo "s_barrier lines in engine_asm.inc: $(grep -c s_barrier engine_asm.inc)"
cd $root
echo "no s_barrier in the k-loops"; python3 tools/lab15.py
echo "no s_barrier + refills from L2 (NOSTEP)"; GPRHIP_LAB_NOSTEP=1 python3 tools/lab15.py
cd gpr_amd/csrc
python3 gen_engine_asm.py engine_asm.inc && make -j8 > /dev/null 2>&1
