#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r3f; rm -f gpurun_out/r3f/*
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_trunk.py -q -m gpu -x 2>&1 | tail -4
timeout 600 python tools/fwd_layer_times.py wgrad 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3f/wgrad_times.txt | tail -4
