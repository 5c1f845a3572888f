#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r3_b; mkdir -p $OUT
python tools/probes/wgrad_bigP.py 2>&1 | grep -v amdgpu.ids | tee $OUT/wgrad_bigP.txt
timeout 1800 python -m pytest tests/test_gpu_dist_two_ranks.py tests/test_gpu_train_step.py "tests/test_gpu_parity_full.py::test_slowfast_r50_one_clip_224_eval_logits_fp32_residual_stream" -q -m gpu --no-header -p no:cacheprovider -s > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest.log
grep -E "relative|decomposition|passed|failed|FAILED|Error|assert" $OUT/pytest.log | cut -c1-600 | head -40
