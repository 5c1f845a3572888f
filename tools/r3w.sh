#!/bin/bash
# gradient-fill A/B + tests
export TMPDIR=/tmp
OUT=gpurun_out/r3w; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_dist_nccl.py -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -5 $OUT/pytest.log
