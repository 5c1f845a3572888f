#!/bin/bash
# Round-4: weight-gradient ring kernel with half stages (4 x 32 positions in 64 KiB, 3 steps in flight; VS_WGRAD_HALF=1)
# against the two-stage ring: parity tests with it on, per-layer table, 8 clips.
export TMPDIR=/tmp
OUT=gpurun_out/r4_wgrad_half; mkdir -p $OUT
VS_WGRAD_HALF=1 timeout 900 python -m pytest tests/test_gpu_conv.py -q -m gpu -x -k "wgrad" --no-header -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest half exit $?"; tail -3 $OUT/pytest.log
for h in 1 0 1 0; do
  VS_WGRAD_HALF=$h timeout 900 python tools/fwd_layer_times.py wgrad > $OUT/wgrad_8_half${h}_$RANDOM.txt 2>&1; echo "half $h: $(tail -1 $(ls -t $OUT/wgrad_8_half${h}_*.txt | head -1))"
done
