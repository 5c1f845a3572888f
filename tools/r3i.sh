#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r3i; rm -f gpurun_out/r3i/*
for ff in 0 1 0 1; do
  echo "== VS_RING_FRAGS_FIRST=$ff"
  VS_RING_FRAGS_FIRST=$ff timeout 600 python tools/fwd_layer_times.py fwd dgrad 2>&1 | grep -v amdgpu.ids | grep -E "^fwd:|^dgrad:|s4.p0.b0.a|s5.p0.b0.a|s2.p0.b0.b|s3.p0.b1.b |s4.p0.b1.a|s3.p0.b0.b"
done | tee gpurun_out/r3i/ab.txt
