#!/bin/bash
# Round-4: the one-thread-per-column slab reduce (S <= 16): weight-gradient tests (bitwise vs the batched slice-form kernel), then the
# step A/B against the previous build (tools/r4_ab_lib.sh).
export TMPDIR=/tmp
OUT=gpurun_out/r4_reduce; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_conv.py tests/test_gpu_trunk.py tests/test_gpu_train_step.py -q -m gpu --no-header -p no:cacheprovider -x -k "wgrad or reduce or grad or step or pair" > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest.log; tail -4 $OUT/pytest.log
bash tools/r4_ab_lib.sh $1 8
