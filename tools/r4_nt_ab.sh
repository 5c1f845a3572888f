#!/bin/bash
# non-temporal hints: optimizer / train-step / txenc tests, then the step A/B against the previous build
export TMPDIR=/tmp
OUT=gpurun_out/r4_nt; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_txenc.py tests/test_gpu_train_step.py tests/test_gpu_trunk.py tests/test_gpu_bn_pool.py -q -m gpu --no-header -p no:cacheprovider -x > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest.log; tail -3 $OUT/pytest.log
bash tools/r4_ab_lib.sh $1 8
