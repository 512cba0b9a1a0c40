#!/bin/bash
# Round 4: what does the per-stage workgroup barrier of the engine loop cost?  Rebuilds the library ON THE GPU BOX with
# GEN_NOBARRIER=1 (the generated loops without their s_barrier: timing only, the results are wrong), times the stages of
# the headline shape, and restores the normal build -- on ANY exit (trap), and loudly if the restore itself fails: a
# no-barrier library left in the tree would give every later test and bench silently wrong numbers.
set -u
root=$(pwd)
restore() {
  cd "$root/gpr_amd/csrc" || { echo "lab16: RESTORE FAILED (cannot enter gpr_amd/csrc): gpr_amd/libgprhip.so may be the NO-BARRIER build" >&2; exit 3; }
  if ! (python3 gen_engine_asm.py engine_asm.inc && make -j8 > "$root/gpurun_out/lab16_restore.log" 2>&1 &&
        [ "$(grep -c s_barrier engine_asm.inc)" -gt 0 ]); then
    echo "lab16: RESTORE BUILD FAILED -- gpr_amd/libgprhip.so is (or may be) the NO-BARRIER build whose results are wrong;" \
         "see gpurun_out/lab16_restore.log and rebuild with: make -C gpr_amd/csrc clean all" >&2
    rm -f "$root/gpr_amd/libgprhip.so"   # better no library than a wrong one
    exit 3
  fi
  echo "lab16: normal build restored ($(grep -c s_barrier engine_asm.inc) s_barrier lines)"
}
trap restore EXIT
mkdir -p gpurun_out
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
echo "no s_barrier + refills from L2 (NOSTEP)"; GPRHIP_LAB=1 GPRHIP_LAB_NOSTEP=1 python3 tools/lab15.py
