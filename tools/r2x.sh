#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r2x; rm -f gpurun_out/r2x/*
for w in 0 1 2 3; do
  VS_WHATIF=$w timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r2x/train_w$w.json 2> gpurun_out/r2x/train_w$w.err
done
grep -o '"ms_per_step": [0-9.]*' gpurun_out/r2x/*.json
