#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r2c; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_conv.py tests/test_gpu_parity_full.py tests/test_gpu_dist_nccl.py tests/test_gpu_bn_pool.py -q -m gpu -s --no-header -p no:cacheprovider -x > $OUT/pytest_a.log 2>&1; echo "pytest a exit $?"; grep -E "passed|failed|rel_l2 |logits|^  [0-9]\.[0-9]+e|Error|error" $OUT/pytest_a.log | tail -60
timeout 900 python -m pytest tests/test_gpu_trunk.py tests/test_gpu_nonlocal.py tests/test_gpu_checkpoint.py -q -m gpu --no-header -p no:cacheprovider -x > $OUT/pytest_b.log 2>&1; echo "pytest b exit $?"; tail -5 $OUT/pytest_b.log
for cfg in "1 1" "0 1" "1 0" "0 0"; do
  set -- $cfg
  VS_MASK_BY_BITS=$1 VS_ACC_SHORTCUT=$2 timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 5 > $OUT/train_m$1_a$2.json 2> $OUT/train_m$1_a$2.err; echo "mask $1 acc $2 exit $?"; head -c 200 $OUT/train_m$1_a$2.json; echo
done
timeout 300 python tools/tile_ab.py > $OUT/tile_ab.txt 2>&1; cat $OUT/tile_ab.txt
timeout 200 python -c "
import bench, json
print(json.dumps(bench.cpu_baseline('sf_txenc_train', 1564)))" > $OUT/cpu_baseline.log 2>&1; tail -2 $OUT/cpu_baseline.log
