#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r2u; rm -f gpurun_out/r2u/*
for cfg in "VS_DIRECT_TB=2" "VS_DIRECT_TB=4"; do
  echo "== $cfg" >> gpurun_out/r2u/log.txt
  env $cfg VS_CONV_PW=0 timeout 600 python -m pytest tests/test_gpu_trunk.py -q -m gpu -x -k "train_step_matches_oracle and slowfast" 2>&1 | tail -60 >> gpurun_out/r2u/log.txt
done
