#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_trunk.py -q -m gpu -x 2>&1 | tail -3
timeout 300 python tools/fwd_layer_times.py fwd dgrad --small 2>&1 | grep -v amdgpu.ids
