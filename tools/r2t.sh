#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r2t; rm -f gpurun_out/r2t/*
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_trunk.py -q -m gpu -x 2>&1 | tail -5 > gpurun_out/r2t/pytest.log
cat gpurun_out/r2t/pytest.log
for pw in 1 0; do
  VS_CONV_PW=$pw timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r2t/train_pw$pw.json 2> gpurun_out/r2t/train_pw$pw.err
  VS_CONV_PW=$pw timeout 300 python bench.py --workload feat_fwd --steps 50 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r2t/fwd_pw$pw.json 2> gpurun_out/r2t/fwd_pw$pw.err
done
grep -o '"ms_per_step": [0-9.]*' gpurun_out/r2t/*.json
