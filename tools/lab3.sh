#!/bin/bash
# lab: paired column tiles (order 3) -- correctness, speed, fabric traffic
set -u
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
for o in 2 3; do echo "== ORD=$o"; WHERE=1 RP=1 ORD=$o timeout 300 ./build/gemm_check | grep -E "checks failed|FAIL|col tile" | head -12; done
echo "== persistent"; GPRHIP_PERSISTENT=1 RP=1 ORD=2 timeout 300 ./build/gemm_check | grep -E "checks failed|FAIL" | head -12
for o in 2 3; do echo "== ORD=$o"; ORD=$o timeout 300 ./build/gemm_check | grep -E "checks failed|FAIL|triu|bad" | head -12; done
echo "== f32"; ORD=3 F32=1 timeout 300 ./build/gemm_check | grep -E "failed|FAIL|triu" | head
cd /tmp && export TMPDIR=/tmp
for o in 2 3; do
  export GPRHIP_TILE_ORDER=$o
  python3 $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('order $o', d['ms_per_step'], d.get('stage_ms'))"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/lab3_$o -- python3 $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-configs > $out/lab3_$o.log 2>&1
  db=$(ls $out/lab3_$o/*/*.db | head -1)
  python3 $root/tools/pmc_summary.py $db $db $db | grep -E "gemm_f64_n" | head -4
  rm -rf $out/lab3_$o
done
