#!/bin/bash
# Round-4: the in-launch split-K plan of the 128 x 128 tile kernel: its tests, then per-layer times of the slow pathway at
# 8 clips with the plan off / heuristic / forced, with and without the halo-image and deep kernels in front of it.
export TMPDIR=/tmp
OUT=gpurun_out/r4_splitk_il; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_conv.py -q -m gpu --no-header -p no:cacheprovider -x -k "splitk or deep_pipeline or halo_image" > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest.log; tail -8 $OUT/pytest.log
ONLY=--only=s3.p0,s4.p0,s5.p0,_fuse
for il in 0 1 2; do
  VS_CONV_SPLITK_IL=$il timeout 600 python tools/fwd_layer_times.py fwd dgrad --clips=8 $ONLY > $OUT/plan_il$il.txt 2>&1
  echo "== default plan order, VS_CONV_SPLITK_IL=$il"; tail -32 $OUT/plan_il$il.txt | cut -c1-120
done
for il in 0 2; do
  VS_CONV_SPLITK_IL=$il timeout 600 python tools/fwd_layer_times.py fwd dgrad --clips=8 --nohalo --nodeep $ONLY > $OUT/tile_il$il.txt 2>&1
  echo "== tile kernel only (--nohalo --nodeep), VS_CONV_SPLITK_IL=$il"; tail -32 $OUT/tile_il$il.txt | cut -c1-120
done
