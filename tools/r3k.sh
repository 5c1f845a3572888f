#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r3k; rm -f gpurun_out/r3k/*
for dbg in 0 1 2 3 64 65 66 67 7 15 79; do
  echo "== VS_WGRAD_DBG=$dbg"
  VS_WGRAD_DBG=$dbg timeout 600 python tools/fwd_layer_times.py wgrad 2>&1 | grep -v amdgpu.ids | grep -E "s4.p0.b0.a|s2.p0.b0.b|s4.p0.b1.b|s5.p0.b1.b|s3.p0.b1.b " | awk '{printf "%s %s  ", $1, $6} END {print ""}'
done | tee gpurun_out/r3k/ab.txt
