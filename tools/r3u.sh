#!/bin/bash
cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_conv.py tests/test_gpu_trunk.py -q -m gpu -x 2>&1 | tail -3
timeout 300 python tools/pw_ab.py 2>&1 | grep -v "amdgpu.ids\|^tiles"
for rep in 1 2; do timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2> /dev/null | grep -o '"ms_per_step": [0-9.]*'; timeout 300 python bench.py --workload feat_fwd --steps 50 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; done
