#!/bin/bash
# persistent pointwise kernel: parity + A/B
cd /root/repo; mkdir -p gpurun_out/r2r; rm -f gpurun_out/r2r/*
timeout 900 python -m pytest tests/test_gpu_conv.py -q -m gpu -x -k "pointwise or pw_ or halo" 2>&1 | tail -5 > gpurun_out/r2r/pytest.log
cat gpurun_out/r2r/pytest.log
for cfg in "X=1" "VS_PW_OCC=1" "VS_PW_BN=64" "VS_PW_BN=128" "VS_PW_BN=64 VS_PW_OCC=1" "VS_PW_BN=128 VS_PW_OCC=1" "VS_PW_NSLOT=3" "VS_PW_NSLOT=5"; do
  echo "== $cfg" >> gpurun_out/r2r/pw_ab.txt
  env $cfg timeout 300 python tools/pw_ab.py 2>&1 | grep -v "amdgpu.ids\|^tiles" >> gpurun_out/r2r/pw_ab.txt
done
cat gpurun_out/r2r/pw_ab.txt
