#!/bin/bash
# Round 3: the apply kernels that finalize for themselves -- parity tests, then an alternating A/B of the train step.
export TMPDIR=/tmp
OUT=gpurun_out/r3_bnfin; mkdir -p $OUT
timeout 2400 python -m pytest tests/test_gpu_bn_pool.py tests/test_gpu_trunk.py tests/test_gpu_main_dist.py tests/test_gpu_train_step.py -q -m gpu --no-header -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest.log; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest.log | cut -c1-300 | tail -20
for i in 1 2 3; do
  for v in 0 1; do
    env VS_BN_FIN_FUSE=$v timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('VS_BN_FIN_FUSE=$v', d['value'], d['ms_per_step'])" | tee -a $OUT/ab.log
  done
done
for v in 0 1; do
  env VS_BN_FIN_FUSE=$v timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 200 --workload feat_fwd 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fwd VS_BN_FIN_FUSE=$v', d['value'], d['ms_per_step'])" | tee -a $OUT/ab.log
done
