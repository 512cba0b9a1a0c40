#!/bin/bash
out=gpurun_out/r02a; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1
tail -5 $out/pytest.txt
python bench.py > $out/bench.json 2> $out/bench.err
tail -3 $out/bench.err
bash tools/profile_round.sh r02 > $out/profile.log 2>&1
tail -12 $out/profile.log
