#!/bin/bash
# Round-4: deep weight-gradient kernel, staggered all-wave copy issue (VS_WGRAD_DEEP_STAG=1) against the halves issue.
export TMPDIR=/tmp
OUT=gpurun_out/r4_wgrad_stag; mkdir -p $OUT
VS_WGRAD_DEEP_STAG=1 timeout 600 python -m pytest tests/test_gpu_conv.py -q -m gpu -x -k "deep_pipeline_wgrad" --no-header -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest stag exit $?"; tail -3 $OUT/pytest.log
for c in 32 8; do for st in 1 0; do
  VS_WGRAD_DEEP_STAG=$st timeout 900 python tools/fwd_layer_times.py wgrad --clips=$c --only=s3.p0,s4.p0,s5.p0 --deep > $OUT/wgrad_${c}_stag$st.txt 2>&1; echo "clips $c stag $st: $(tail -1 $OUT/wgrad_${c}_stag$st.txt)"
done; done
