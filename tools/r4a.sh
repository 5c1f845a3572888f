#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r4a; mkdir -p $OUT
for i in 1 2; do
  for f in 0 1; do
    VS_TXENC_STACK=$f timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 200 2>$OUT/err_$f.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stack=$f', d['value'], d['ms_per_step'])" | tee -a $OUT/ab.log
  done
done
tail -3 $OUT/err_1.log
timeout 900 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_dist_nccl.py tests/test_gpu_trunk.py -q -p no:cacheprovider -k "train_step or nccl or bitwise or overwritten or fill" > $OUT/pytest.log 2>&1; echo "pytest exit $?"; grep -E "AssertionError|Error|passed|failed" $OUT/pytest.log | cut -c1-400 | tail
