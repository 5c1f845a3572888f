#!/bin/bash
# Round-4: full GPU suite on the build with the deep-pipeline kernel, then the train step and the eval forward with the
# kernel's plan on / off (VS_CONV_DEEP=1 / 0), alternating in one session; 8 and 32 clips per GPU.
export TMPDIR=/tmp
OUT=gpurun_out/r4_ab_deep; mkdir -p $OUT
python - <<'PY' > $OUT/visible_gpus.txt 2>&1
from vidsitu_amd import dist_launch
import glob
print("kfd nodes:", len(glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")), "visible_gpus:", dist_launch.visible_gpus())
PY
cat $OUT/visible_gpus.txt
timeout 3000 python -m pytest tests -q -m gpu --no-header -p no:cacheprovider -x > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log; tail -5 $OUT/pytest_gpu.log
cp gpurun_out/parity_eval.json $OUT/ 2>/dev/null
for rep in 1 2; do
for deep in 1 0; do
for c in 8 32; do
  VS_CONV_DEEP=$deep timeout 600 python bench.py --workload sf_txenc_train --clips-per-gpu $c --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-feat-fwd > $OUT/train_${c}_deep${deep}_$rep.json 2> $OUT/train_${c}_deep${deep}_$rep.err
  VS_CONV_DEEP=$deep timeout 600 python bench.py --workload feat_fwd --clips-per-gpu $c --steps 50 --warmup 5 --no-cpu-baseline --no-roofline > $OUT/fwd_${c}_deep${deep}_$rep.json 2> $OUT/fwd_${c}_deep${deep}_$rep.err
  python - <<PY
import json
for wl in ("train","fwd"):
    try:
        d=json.loads(open("$OUT/%s_${c}_deep${deep}_$rep.json"%wl).read().strip().splitlines()[-1]); print(wl,"clips",$c,"deep",$deep,"rep",$rep,d["value"],d["ms_per_step"])
    except Exception as e: print(wl,"no line",e)
PY
done; done; done
