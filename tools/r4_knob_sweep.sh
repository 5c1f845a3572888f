#!/bin/bash
# Round-4, late build: every shipped switch / plan knob alone against the default in the train step, alternating, one GPU session
export TMPDIR=/tmp
OUT=gpurun_out/r4_knobs; mkdir -p $OUT; F=$OUT/knobs.txt; : > $F
run() { env $1 timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-feat-fwd --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-30s %8.2f clips/s %7.3f ms' % ('$1', d['value'], d['ms_per_step']))" | tee -a $F; }
for i in 1 2; do
  for cfg in DEFAULT=1 VS_CONV_DEEP=2 VS_CONV_DEEP=0 VS_CONV_KORDER_MIN=0 VS_WGRAD_DEEP_STAG=0 VS_WGRAD_ALIGN8=1 VS_WGRAD_DEEP_ALIGN=1 VS_TRAIN_AOL=1 \
             DEFAULT=2 VS_DIRECT_BNB=1 VS_WGRAD_XCD=0 VS_CONV_HALO=2 VS_CONV_HALO=0 VS_HALO_CONFLICT_WEIGHT=0.6 VS_REDUCE_MERGE=0 VS_CONV_PAIR=0 \
             DEFAULT=3 VS_BN_TWO_LEVEL=1600 VS_BN_TWO_LEVEL=256 VS_WGRAD_LANES=0 VS_ADAM_OVERLAP=1 VS_WGRAD_SLOTS_SMALL=256 VS_WGRAD_SLOTS_SMALL=1024 VS_DIRECT_TB=2 VS_WGRAD_HALF=1; do run $cfg; done
done
