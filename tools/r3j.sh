#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r3j; rm -f gpurun_out/r3j/*
timeout 900 python -m pytest tests/test_gpu_conv.py -q -m gpu -x 2>&1 | tail -2
for ff in 0 1 0 1; do
  echo "== VS_WGRAD_FRAGS_FIRST=$ff"
  VS_WGRAD_FRAGS_FIRST=$ff timeout 600 python tools/fwd_layer_times.py wgrad 2>&1 | grep -v amdgpu.ids | grep -E "^wgrad:|s4.p0.b0.a|s5.p0.b0.a|s2.p0.b0.b|s3.p0.b1.b |s4.p0.b1.a|s4.p0.b1.b|s5.p0.b1.b|s4.p0.b0.c|s2.p0.b0.c"
done | tee gpurun_out/r3j/ab.txt
