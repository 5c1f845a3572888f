#!/bin/bash
export TMPDIR=/tmp; OUT=gpurun_out/ff; mkdir -p $OUT
run() { env $1 $2 timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 150 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-46s %8.2f clips/s %7.3f ms' % ('$1 $2', d['value'], d['ms_per_step']))" | tee -a $OUT/sweep.txt; }
for i in 1 2; do
  run VS_RING_FF_MIN=20 VS_RING_FF_MAX=100000
  run VS_RING_FF_MIN=100000 VS_RING_FF_MAX=100000
  run VS_RING_FF_MIN=0 VS_RING_FF_MAX=100000
  run VS_RING_FF_MIN=0 VS_RING_FF_MAX=19
  run VS_RING_FF_MIN=0 VS_RING_FF_MAX=8
  run VS_RING_FF_MIN=9 VS_RING_FF_MAX=19
  run VS_RING_FF_MIN=20 VS_RING_FF_MAX=40
  run VS_RING_FF_MIN=41 VS_RING_FF_MAX=100000
done
