#!/bin/bash
out=gpurun_out/lab11; mkdir -p $out
timeout 120 ./build/gemm_check > $out/check.txt 2>&1; echo "rc=$?" >> $out/check.txt
F32=1 timeout 120 ./build/gemm_check > $out/check_f32.txt 2>&1; echo "rc=$?" >> $out/check_f32.txt
LAB=1 timeout 120 ./build/gemm_check > $out/lab.txt 2>&1
python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/bench.json 2> $out/bench.err
grep -v "^check.*OK" $out/check.txt $out/check_f32.txt; tail -9 $out/lab.txt; tail -5 $out/pytest.txt; python - <<'PY'
import json
d=json.loads(open("gpurun_out/lab11/bench.json").read()); print(d["ms_per_step"], d["stage_ms"])
PY
