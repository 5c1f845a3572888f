#!/bin/bash
cd /root/repo
for rep in 1 2; do for cfg in "VS_DUAL_STREAM=1" "VS_DUAL_STREAM=0"; do
  env $cfg timeout 300 python bench.py --workload feat_fwd --steps 50 --warmup 5 --no-cpu-baseline --no-roofline 2> /dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/fwd $cfg rep$rep /"
done; done
