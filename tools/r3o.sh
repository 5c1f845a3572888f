#!/bin/bash
cd /root/repo
for rep in 1 2; do for cfg in "VS_BN_TWO_LEVEL=512" "VS_BN_TWO_LEVEL=256" "VS_BN_TWO_LEVEL=128" "VS_BN_TWO_LEVEL=64"; do
  env $cfg timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2> /dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/$cfg rep$rep /"
done; done
