#!/bin/bash
# in-step sweep of the numeric plan knobs (each alone against the default), alternating, one GPU session
export TMPDIR=/tmp
OUT=gpurun_out/ablation; mkdir -p $OUT; F=$OUT/knobs.txt; : > $F
run() { env $1 timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 150 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s %8.2f clips/s %7.3f ms' % ('$1', d['value'], d['ms_per_step']))" | tee -a $F; }
for i in 1 2; do
  for cfg in DEFAULT=1 VS_WGRAD_SLOTS=256 VS_WGRAD_SLOTS=512 VS_WGRAD_SLOTS_SMALL=256 VS_WGRAD_SLOTS_SMALL=1024 VS_WGRAD_XCD=0 VS_DIRECT_BNB=1 \
             VS_BN_TARGET=1024 VS_BN_TARGET=4096 VS_BN_TWO_LEVEL=1600 VS_DIRECT_TB=2; do run $cfg; done
done
