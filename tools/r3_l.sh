#!/bin/bash
mkdir -p gpurun_out/r3_l
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_trunk.py -q -m gpu -k "fused_bc" -x 2>&1 | tail -5
for v in 1 0 1 0; do
  VS_EVAL_FUSE_BC=$v timeout 600 python bench.py --workload feat_fwd --steps 50 --warmup 10 --no-cpu-baseline --no-roofline \
    2> gpurun_out/r3_l/err_$v.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fuse_bc=$v', d['value'], d['ms_per_step'])"
done
