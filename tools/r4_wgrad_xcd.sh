#!/bin/bash
# Round-4: weight gradients with one position split per XCD (S a multiple of 8 + XCD-contiguous block order) against
# the round-3 split choice (VS_WGRAD_ALIGN8=0), per layer and in the step; then the train step's timeline.
export TMPDIR=/tmp
OUT=gpurun_out/r4_wgrad_xcd; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_conv.py -q -m gpu -x -k "wgrad" --no-header -p no:cacheprovider > $OUT/pytest_wgrad.log 2>&1; echo "pytest wgrad exit $?"; tail -3 $OUT/pytest_wgrad.log
for a in 1 0; do
  VS_WGRAD_ALIGN8=$a timeout 900 python tools/fwd_layer_times.py wgrad > $OUT/wgrad_8_align$a.txt 2>&1; echo "wgrad table align $a exit $?"; tail -1 $OUT/wgrad_8_align$a.txt
done
VS_WGRAD_ALIGN8=1 VS_WGRAD_XCD=2 timeout 900 python tools/fwd_layer_times.py wgrad > $OUT/wgrad_8_align1_xcd2.txt 2>&1; tail -1 $OUT/wgrad_8_align1_xcd2.txt
for rep in 1 2 3; do for a in 1 0; do
  VS_WGRAD_ALIGN8=$a timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-feat-fwd 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train align8 $a', d['value'], d['ms_per_step'])"
done; done
bash tools/timeline.sh sf_txenc_train > $OUT/timeline_train.txt 2>&1; tail -70 $OUT/timeline_train.txt
