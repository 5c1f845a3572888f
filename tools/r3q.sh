#!/bin/bash
cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_conv.py tests/test_gpu_trunk.py -q -m gpu -x 2>&1 | tail -3
timeout 600 python tools/fwd_layer_times.py dgrad 2>&1 | grep -v amdgpu.ids | grep -E "fuse|b0.b |b0.sc|^dgrad:"
