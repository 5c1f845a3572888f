#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r3_parity; mkdir -p $OUT
timeout 3000 python -m pytest tests/test_gpu_parity_full.py tests/test_gpu_bn_pool.py tests/test_gpu_train_step.py -q -m gpu --no-header -p no:cacheprovider -s > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest.log
grep -E "relative|rel_l2|passed|failed|FAILED|Error" $OUT/pytest.log | cut -c1-400 | head -60
for v in 0 1; do
  env VS_RESIDUAL_FP32=$v timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 100 --workload feat_fwd 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fwd VS_RESIDUAL_FP32=$v', d['value'], d['ms_per_step'])" | tee -a $OUT/ab.log
done
