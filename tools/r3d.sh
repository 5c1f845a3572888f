#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r3d; rm -f gpurun_out/r3d/*
for rep in 1 2; do
for cfg in "VS_ADAM_OVERLAP=1" "VS_ADAM_OVERLAP=2" "VS_ADAM_OVERLAP=0"; do
  env $cfg timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2> /dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/$cfg rep$rep /"
done
done | tee gpurun_out/r3d/ab.txt
