#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r3b; rm -f gpurun_out/r3b/*
for rep in 1 2; do
for s in 512 1024 2048; do
  VS_WGRAD_SLOTS_SMALL=$s timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2> /dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/small_slots=$s rep$rep /"
done
done | tee gpurun_out/r3b/ab.txt
