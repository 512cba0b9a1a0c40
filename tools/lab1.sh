#!/bin/bash
root=$(pwd); out=$root/gpurun_out/lab12; mkdir -p $out
for o in 0 1 2; do echo ORD=$o; ORD=$o LAB=1 ./build/gemm_check | grep "triu\|full, K=2048" | head -3; done
cd /tmp && export TMPDIR=/tmp
for o in 0 2; do
ORD=$o LAB=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/f_o$o -- $root/build/gemm_check > /dev/null 2>&1
done
cd $root; python3 - <<'PY'
import csv,glob,collections
for o in (0,2):
    f=glob.glob('gpurun_out/lab12/f_o%d/*/*counter_collection.csv'%o)[0]
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name']=='FETCH_SIZE': agg[(r['Kernel_Name'][:30],r['Grid_Size'])].append(float(r['Counter_Value']))
    for k,v in agg.items(): print(o,k,len(v),'%.2f GB (x2 corrected)'%(2*sum(v)/len(v)*1024/1e9))
PY
rm -rf $out/f_o*
