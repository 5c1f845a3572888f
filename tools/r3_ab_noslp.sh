#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r3_d; mkdir -p $OUT
export VS_WGRAD_S1_TILES=128
for i in 1 2 3; do
  for lib in "" "/root/repo/tmp/noslp/libvidsitu_hip.so"; do
    env VS_LIB_PATH=$lib timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train lib=${lib:-default}', d['value'], d['ms_per_step'])" | tee -a $OUT/noslp_ab.log
  done
done
for lib in "" "/root/repo/tmp/noslp/libvidsitu_hip.so"; do
  env VS_LIB_PATH=$lib timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 200 --workload feat_fwd 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fwd lib=${lib:-default}', d['value'], d['ms_per_step'])" | tee -a $OUT/noslp_ab.log
done
