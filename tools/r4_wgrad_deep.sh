#!/bin/bash
# Round-4: the deep-pipeline weight-gradient kernel -- parity tests, per-layer table forced / plan / off at 8 and 32 clips,
# the train step with the plan on / off (VS_WGRAD_DEEP=1 / 0).
export TMPDIR=/tmp
OUT=gpurun_out/r4_wgrad_deep; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_conv.py -q -m gpu -x -k "wgrad" --no-header -p no:cacheprovider > $OUT/pytest_wgrad.log 2>&1; echo "pytest wgrad exit $?"; tail -12 $OUT/pytest_wgrad.log
for c in 8 32; do
  only=""; [ $c = 32 ] && only="--only=s2.p0,s3.p0,s4.p0,s5.p0"
  timeout 900 python tools/fwd_layer_times.py wgrad --clips=$c $only --deep > $OUT/wgrad_${c}_force.txt 2>&1; echo "force $c exit $?"; tail -1 $OUT/wgrad_${c}_force.txt
  timeout 900 python tools/fwd_layer_times.py wgrad --clips=$c $only > $OUT/wgrad_${c}_plan.txt 2>&1; echo "plan $c exit $?"; tail -1 $OUT/wgrad_${c}_plan.txt
  timeout 900 python tools/fwd_layer_times.py wgrad --clips=$c $only --nodeep > $OUT/wgrad_${c}_off.txt 2>&1; echo "off $c exit $?"; tail -1 $OUT/wgrad_${c}_off.txt
done
for rep in 1 2 3; do for a in 1 0; do
  VS_WGRAD_DEEP=$a timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-feat-fwd 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train wgrad_deep $a', d['value'], d['ms_per_step'])"
done; done
