#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r2g; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_conv.py tests/test_gpu_trunk.py tests/test_gpu_parity_full.py tests/test_gpu_dist_nccl.py tests/test_gpu_checkpoint.py tests/test_gpu_nonlocal.py -q -m gpu --no-header -p no:cacheprovider > $OUT/pytest_a.log 2>&1; echo "pytest a exit $?"; tail -6 $OUT/pytest_a.log
for b in 1 0; do
VS_WGRAD_BATCH=$b timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 5 > $OUT/train_batch$b.json 2> $OUT/train_batch$b.err; echo "batch $b exit $?"; head -c 200 $OUT/train_batch$b.json; echo
done
VS_WGRAD_BATCH=1 VS_WGRAD_SLOTS=256 timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 5 > $OUT/train_batch_s256.json 2> /dev/null; head -c 200 $OUT/train_batch_s256.json; echo
VS_WGRAD_BATCH=1 VS_WGRAD_SLOTS=512 timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 5 > $OUT/train_batch_s512.json 2> /dev/null; head -c 200 $OUT/train_batch_s512.json; echo
