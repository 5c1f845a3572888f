#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r2e; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_conv.py -q -m gpu -k "halo" --no-header -p no:cacheprovider > $OUT/pytest_halo.log 2>&1; echo "pytest halo exit $?"; tail -15 $OUT/pytest_halo.log
timeout 1200 python -m pytest tests/test_gpu_conv.py tests/test_gpu_parity_full.py tests/test_gpu_dist_nccl.py tests/test_gpu_checkpoint.py tests/test_gpu_trunk.py -q -m gpu --no-header -p no:cacheprovider > $OUT/pytest_a.log 2>&1; echo "pytest a exit $?"; tail -8 $OUT/pytest_a.log
timeout 600 python tools/fwd_layer_times.py fwd dgrad > $OUT/layer_times.txt 2>&1; echo "layers exit $?"; head -28 $OUT/layer_times.txt; tail -3 $OUT/layer_times.txt
timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 5 > $OUT/train.json 2> $OUT/train.err; echo "train exit $?"; head -c 200 $OUT/train.json; echo
VS_CONV_HALO=0 timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 5 > $OUT/train_nohalo.json 2> $OUT/train_nohalo.err; echo "train nohalo exit $?"; head -c 200 $OUT/train_nohalo.json; echo
timeout 300 python bench.py --workload feat_fwd --no-cpu-baseline --no-roofline --steps 30 --warmup 5 > $OUT/fwd.json 2> $OUT/fwd.err; echo "fwd exit $?"; head -c 200 $OUT/fwd.json; echo
VS_CONV_HALO=0 timeout 300 python bench.py --workload feat_fwd --no-cpu-baseline --no-roofline --steps 30 --warmup 5 > $OUT/fwd_nohalo.json 2> $OUT/fwd_nohalo.err; echo "fwd nohalo exit $?"; head -c 200 $OUT/fwd_nohalo.json; echo
