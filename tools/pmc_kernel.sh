#!/bin/bash
# SQ counter passes for one kernel of a command (name substring in $1; command after --), e.g.
#   bash tools/pmc_kernel.sh grad_mfma -- python3 tools/run_config3.py
# Prints per-launch averages.  (GPU box, repo root; each --pmc group is its own run.)
set -u
pat=$1; shift; shift
root=$(pwd); out=$root/gpurun_out/pmck; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_F64 SQ_WAVES"; do
  i=$((i+1))
  (cd $root && rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/g$i -- "$@" > $out/g$i.log 2>&1)
done
cd $root
python3 - "$pat" <<'PY'
import csv,glob,collections,sys
pat=sys.argv[1]
agg=collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmck/g*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()): print("%-28s n=%3d avg %.4g"%(k,len(v),sum(v)/len(v)))
PY
rm -rf $out
