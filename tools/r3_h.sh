#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r3_h; mkdir -p $OUT
for cfg in "A=0" "VS_WGRAD_LANE_MIN_GFLOP=10" "VS_WGRAD_LANE_MIN_GFLOP=20" "VS_WGRAD_LANE_MIN_GFLOP=50" "A=0" "VS_WGRAD_LANE_MIN_GFLOP=100" "VS_BN_TWO_LEVEL=256" "VS_BN_TWO_LEVEL=1024" "A=0" "VS_WGRAD_LANE_MIN_GFLOP=20" "VS_WGRAD_LANE_MIN_GFLOP=50"; do
  env $cfg timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['value'], d['ms_per_step'])" | tee -a $OUT/sweep.log
done
