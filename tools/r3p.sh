#!/bin/bash
cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_conv.py tests/test_gpu_trunk.py tests/test_gpu_parity_full.py tests/test_gpu_train_step.py -q -m gpu -x 2>&1 | tail -4
for rep in 1 2 3; do for cfg in "VS_FUSE_SC_SUMS=1" "VS_FUSE_SC_SUMS=0"; do
  env $cfg timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2> /dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/$cfg rep$rep /"
done; done
