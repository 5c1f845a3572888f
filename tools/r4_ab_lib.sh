#!/bin/bash
# A/B of two builds of the library in one session (alternating): tools/r4_ab_lib.sh <other .so> [clips ...]
# The train step and the eval forward with the in-tree library ("new") and with VS_LIB_PATH=<other> ("old").
export TMPDIR=/tmp
OTHER=$1; shift; CL=${@:-8}
OUT=gpurun_out/r4_ab_lib; mkdir -p $OUT
for rep in 1 2 3; do
for which in new old; do
for c in $CL; do
  if [ $which = old ]; then export VS_LIB_PATH=$OTHER; else unset VS_LIB_PATH; fi
  timeout 600 python bench.py --workload sf_txenc_train --clips-per-gpu $c --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-feat-fwd > $OUT/train_${c}_${which}_$rep.json 2> $OUT/train_${c}_${which}_$rep.err
  timeout 600 python bench.py --workload feat_fwd --clips-per-gpu $c --steps 50 --warmup 5 --no-cpu-baseline --no-roofline > $OUT/fwd_${c}_${which}_$rep.json 2> $OUT/fwd_${c}_${which}_$rep.err
  python - <<PY
import json
for wl in ("train","fwd"):
    try:
        d=json.loads(open("$OUT/%s_${c}_${which}_$rep.json"%wl).read().strip().splitlines()[-1]); print(wl,"clips",$c,"$which","rep",$rep,d["value"],d["ms_per_step"])
    except Exception as e: print(wl,"no line",e)
PY
done; done; done
