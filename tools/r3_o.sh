#!/bin/bash
mkdir -p gpurun_out/r3_o
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_trunk.py -q -m gpu -k "apply_on_load" -x 2>&1 | tail -5
run() { echo "$*"; env "$@" timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2> gpurun_out/r3_o/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   ', d['value'], d['ms_per_step'])" || tail -5 gpurun_out/r3_o/err.txt; }
for rep in 1 2 3; do
run VS_TRAIN_AOL=0
run VS_TRAIN_AOL=1
done
