#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r2z; rm -f gpurun_out/r2z/*
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r2z/train.json 2> gpurun_out/r2z/train.err
timeout 300 python bench.py --workload feat_fwd --steps 50 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r2z/fwd.json 2> gpurun_out/r2z/fwd.err
grep -o '"ms_per_step": [0-9.]*' gpurun_out/r2z/*.json
timeout 900 python -m pytest tests/test_gpu_trunk.py tests/test_gpu_parity_full.py -q -m gpu -x 2>&1 | tail -3
