#!/bin/bash
cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_txenc.py tests/test_gpu_txdec.py tests/test_gpu_train_step.py tests/test_gpu_dist_nccl.py tests/test_gpu_checkpoint.py tests/test_gpu_trunk.py -q -m gpu -x 2>&1 | tail -4
for rep in 1 2; do for cfg in "X=1" "VS_LINEAR_BWD_FUSED=0 VS_LN_BWD_FUSED=0" "VS_LINEAR_BWD_FUSED=0" "VS_LN_BWD_FUSED=0"; do
  env $cfg timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2> /dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/$cfg rep$rep /"
done; done
