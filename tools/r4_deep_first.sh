#!/bin/bash
# Round-4: first GPU session of the deep-pipeline conv kernel -- its parity tests, then the per-layer tables with the
# kernel forced on / off at 8 and 32 clips (slow pathway, MFMA-side layers).
export TMPDIR=/tmp
OUT=gpurun_out/r4_deep_first; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_conv.py -q -m gpu -x -k "deep_pipeline" --no-header -p no:cacheprovider > $OUT/pytest_deep.log 2>&1; echo "pytest deep exit $?"; tail -15 $OUT/pytest_deep.log
for c in 8 32; do
  timeout 600 python tools/fwd_layer_times.py fwd dgrad --clips=$c --only=s3.p0,s4.p0,s5.p0 --deep > $OUT/layers_${c}_deep.txt 2>&1; echo "deep $c exit $?"
  timeout 600 python tools/fwd_layer_times.py fwd dgrad --clips=$c --only=s3.p0,s4.p0,s5.p0 --nodeep > $OUT/layers_${c}_nodeep.txt 2>&1; echo "nodeep $c exit $?"
done
tail -30 $OUT/layers_8_deep.txt | cut -c1-120
