#!/bin/bash
# SQ counter passes over the production shapes (engine geometry from GPRHIP_ENG_WAVES)
root=$(pwd); out=$root/gpurun_out/lab3; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $out/counters.txt 2>&1
for w in 4 8; do
export GPRHIP_ENG_WAVES=$w LAB=1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/p1_w$w -- $root/build/gemm_check > $out/p1_w$w.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS --kernel-trace --output-format csv -d $out/p2_w$w -- $root/build/gemm_check > $out/p2_w$w.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM --kernel-trace --output-format csv -d $out/p3_w$w -- $root/build/gemm_check > $out/p3_w$w.log 2>&1
done
find $out -name "*.csv" | head -30; du -sh $out
