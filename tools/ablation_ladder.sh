#!/bin/bash
# Every shipped switch of a bench workload turned off alone against the default build, one GPU session:
# bash tools/ablation_ladder.sh [reps] [sf_txenc_train|feat_fwd]   -> gpurun_out/ablation/ladder_<workload>.txt
export TMPDIR=/tmp
WL=${2:-sf_txenc_train}
OUT=gpurun_out/ablation; mkdir -p $OUT; F=$OUT/ladder_$WL.txt; : > $F
run() { env $1 timeout 300 python bench.py --workload $WL --no-cpu-baseline --no-roofline --steps 150 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s %8.2f clips/s %7.3f ms' % ('$1', d['value'], d['ms_per_step']))" | tee -a $F; }
if [ $WL = feat_fwd ]; then
  CFGS="DEFAULT=1 VS_DUAL_STREAM=0 VS_CONV_PW=0 VS_CONV_HALO=0 VS_STEM_PAIR=0 VS_DIRECT_TB=1"
else
  CFGS="DEFAULT=1 VS_DUAL_STREAM=0 VS_WGRAD_LANES=0 VS_FUSE_BN_SUMS=0 VS_FUSE_SC_SUMS=0 VS_ACC_SHORTCUT=0 VS_CONV_PW=0 VS_STEM_PAIR=0 VS_STEM_POOL_FUSE=0 VS_GRAD_FILL=1 VS_LINEAR_BWD_FUSED=0 VS_LN_BWD_FUSED=0 VS_RESIDUAL_ROUTE=0 VS_TRANSPOSE_LATE=0 VS_CONV_PAIR=0 VS_BN_FIN2=0 VS_FUSE_SC_APPLY=0 VS_FUSE_SC_BWD=0 VS_WGRAD_GROUP=0 VS_WGRAD_GROUP_SPAN=1 VS_FUSE_ON_FAST=0 VS_DIRECT_TB=1 VS_CONV_HALO=0"
fi
for i in $(seq 1 ${1:-2}); do for cfg in $CFGS; do run $cfg; done; done
