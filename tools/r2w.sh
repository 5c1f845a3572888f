#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r2w; rm -f gpurun_out/r2w/*
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_trunk.py -q -m gpu -x 2>&1 | tail -4 > gpurun_out/r2w/pytest.log
cat gpurun_out/r2w/pytest.log
timeout 300 python tools/pw_ab.py 2>&1 | grep -v "amdgpu.ids\|^tiles" > gpurun_out/r2w/pw_ab.txt
cat gpurun_out/r2w/pw_ab.txt
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r2w/train.json 2> gpurun_out/r2w/train.err
timeout 300 python bench.py --workload feat_fwd --steps 50 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r2w/fwd.json 2> gpurun_out/r2w/fwd.err
grep -o '"ms_per_step": [0-9.]*' gpurun_out/r2w/*.json
