#!/bin/bash
cd /root/repo
for rep in 1 2 3; do for cfg in "VS_WGRAD_SLOTS=384" "VS_WGRAD_SLOTS=320" "VS_WGRAD_SLOTS=256" "VS_WGRAD_SLOTS=192" "VS_WGRAD_SLOTS=128"; do
  env $cfg timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2> /dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/$cfg rep$rep /"
done; done
