#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r2d; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_conv.py tests/test_gpu_parity_full.py tests/test_gpu_dist_nccl.py tests/test_gpu_checkpoint.py -q -m gpu -s --no-header -p no:cacheprovider > $OUT/pytest_a.log 2>&1; echo "pytest a exit $?"; grep -E "passed|failed|^  [0-9]\.[0-9]+e|Error|error" $OUT/pytest_a.log | tail -50
timeout 900 python -m pytest tests/test_gpu_trunk.py tests/test_gpu_nonlocal.py tests/test_gpu_bn_pool.py -q -m gpu --no-header -p no:cacheprovider > $OUT/pytest_b.log 2>&1; echo "pytest b exit $?"; tail -5 $OUT/pytest_b.log
timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 5 > $OUT/train.json 2> $OUT/train.err; echo "train exit $?"; head -c 200 $OUT/train.json; echo
timeout 300 python tools/tile_ab.py > $OUT/tile_ab.txt 2>&1; cat $OUT/tile_ab.txt
