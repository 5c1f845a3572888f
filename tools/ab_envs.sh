#!/bin/bash
# alternating sweep of one environment variable over several values: tools/ab_envs.sh VAR "v1 v2 v3" [reps]
export TMPDIR=/tmp
OUT=gpurun_out/ab_env; mkdir -p $OUT
for i in $(seq 1 ${3:-2}); do
  for v in $2; do
    env $1=$v timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1=$v', d['value'], d['ms_per_step'])" | tee -a $OUT/ab.log
  done
done
