#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r2h; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_conv.py -q -m gpu --no-header -p no:cacheprovider -k "strided or inplace or masked" > $OUT/pytest_a.log 2>&1; echo "pytest a exit $?"; tail -3 $OUT/pytest_a.log
timeout 300 python tools/tile_ab.py > $OUT/tile_ab.txt 2>&1; grep -E "sc dgrad|b0.b dgrad|fuse" $OUT/tile_ab.txt
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 5 > $OUT/train_$i.json 2> $OUT/train_$i.err; echo "train $i exit $?"; head -c 200 $OUT/train_$i.json; echo
done
