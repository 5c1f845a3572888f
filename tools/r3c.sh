#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r3c; rm -f gpurun_out/r3c/*
for rep in 1 2; do
for cfg in "A=0" "VS_WGRAD_BATCH=1" "VS_WGRAD_BATCH=1 VS_WHATIF=4" "VS_WHATIF=2"; do
  env $cfg timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2> /dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/$cfg rep$rep /"
done
done | tee gpurun_out/r3c/ab.txt
