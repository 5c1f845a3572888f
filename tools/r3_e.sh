#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r3_e; mkdir -p $OUT
timeout 2400 python -m pytest tests/test_gpu_trunk.py tests/test_gpu_train_step.py tests/test_gpu_dist_nccl.py tests/test_gpu_dist_two_ranks.py tests/test_gpu_bn_pool.py -q -m gpu --no-header -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest.log; grep -E "^(FAILED|ERROR)|passed|failed|^E  " $OUT/pytest.log | cut -c1-400 | tail -20
for i in 1 2 3; do
  for v in 0 1; do
    env VS_REDUCE_MERGE=$v timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('VS_REDUCE_MERGE=$v', d['value'], d['ms_per_step'])" | tee -a $OUT/ab.log
  done
done
