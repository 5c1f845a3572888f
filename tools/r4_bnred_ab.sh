#!/bin/bash
# BN reduce kernels with predicated 16-load rounds: BN / trunk / train-step tests, then the step A/B against the previous build
export TMPDIR=/tmp
OUT=gpurun_out/r4_bnred; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_bn_pool.py tests/test_gpu_trunk.py tests/test_gpu_train_step.py -q -m gpu --no-header -p no:cacheprovider -x > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest.log; tail -3 $OUT/pytest.log
bash tools/r4_ab_lib.sh $1 8
