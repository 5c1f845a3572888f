#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r4b; mkdir -p $OUT
for i in 1 2; do
  for cfg in "0 0" "1 6" "1 3"; do
    set -- $cfg
    VS_TXENC_STACK=$1 VS_TX_BAR=$2 timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stack=$1 bar=$2', d['value'], d['ms_per_step'])" | tee -a $OUT/ab.log
  done
done
