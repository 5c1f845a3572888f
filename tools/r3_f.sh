#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r3_f; mkdir -p $OUT
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')" 2>&1 | tail -1
for i in 1 2; do
  for v in 0 -1 1; do
    env VS_SIDE_PRIORITY=$v timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('VS_SIDE_PRIORITY=$v', d['value'], d['ms_per_step'])" | tee -a $OUT/prio.log
  done
done
for v in 0 -1 1; do
  env VS_SIDE_PRIORITY=$v timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 200 --workload feat_fwd 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fwd VS_SIDE_PRIORITY=$v', d['value'], d['ms_per_step'])" | tee -a $OUT/prio.log
done
