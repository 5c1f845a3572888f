#!/bin/bash
# Round-3 first GPU session: the new launcher / main_dist / fill-guard tests, the train bench line with the new roofline fields.
export TMPDIR=/tmp
OUT=gpurun_out/r3_first; mkdir -p $OUT
timeout 1800 python -m pytest tests/test_gpu_main_dist.py tests/test_gpu_train_step.py tests/test_gpu_dist_nccl.py -q -m gpu --no-header -p no:cacheprovider -x > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest.log; tail -30 $OUT/pytest.log
( time timeout 900 python bench.py ) > $OUT/bench_train.json 2> $OUT/bench_train.err; echo "bench train exit $?"; head -c 600 $OUT/bench_train.json; echo; tail -4 $OUT/bench_train.err
