#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r2y; rm -f gpurun_out/r2y/*
timeout 900 python -m pytest tests/test_gpu_conv.py -q -m gpu -x -k "stem" 2>&1 | tail -5 > gpurun_out/r2y/pytest.log
cat gpurun_out/r2y/pytest.log
for sp in 1 2 0; do echo "VS_STEM_PAIR=$sp"; VS_STEM_PAIR=$sp timeout 300 python tools/stem_time.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r2y/stem_time.txt
