#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r3f
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_trunk.py -q -m gpu -x 2>&1 | tail -3
timeout 600 python tools/fwd_layer_times.py wgrad 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3f/wgrad_times2.txt | grep -E "s5.p0.b1|s4.p0.b1|s2.p0.b0.b|s4.p0.b0.c|s3.p0.b0.c|s3.p0.b1.b|wgrad:"
