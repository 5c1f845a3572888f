#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r3e; rm -f gpurun_out/r3e/*
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_trunk.py tests/test_gpu_parity_full.py -q -m gpu -x 2>&1 | tail -6
for rep in 1 2; do
for cfg in "VS_DIRECT_BNB=1" "VS_DIRECT_BNB=0"; do
  env $cfg timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2> /dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/$cfg rep$rep /"
done
done | tee gpurun_out/r3e/ab.txt
