#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r3_g; mkdir -p $OUT
for i in 1 2 3; do
  for v in 0 -1; do
    env VS_MAIN_PRIORITY=$v timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('VS_MAIN_PRIORITY=$v', d['value'], d['ms_per_step'])" | tee -a $OUT/prio.log
  done
done
for v in 0 -1; do
  env VS_MAIN_PRIORITY=$v timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 200 --workload feat_fwd 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fwd VS_MAIN_PRIORITY=$v', d['value'], d['ms_per_step'])" | tee -a $OUT/prio.log
done
