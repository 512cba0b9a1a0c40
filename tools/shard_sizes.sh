#!/bin/bash
# Whole-evaluation and per-stage times at the row counts one GPU holds when n=1M is split over 8 / 4 GPUs
# (profiles/rNN_shard_sizes.txt).  usage (GPU box, repo root): bash tools/shard_sizes.sh
set -u
root=$(pwd)
for n in 125000 250000; do
python3 $root/bench.py --points $n --steps 10 --warmup 3 --no-cpu-baseline --no-configs 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print($n, d['ms_per_step'], 'sum_stages', sum(s.values())); print({k:round(v,2) for k,v in s.items()}); print('evidence_only', d['evidence_only']['ms_per_step'])"
done
