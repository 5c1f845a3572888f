#!/bin/bash
# direct (small-channel) kernel: tiles in flight per wave A/B + parity
cd /root/repo; mkdir -p gpurun_out/r2q
timeout 600 python -m pytest tests/test_gpu_conv.py -q -m gpu -x 2>&1 | tail -3 > gpurun_out/r2q/pytest.log
for tb in 1 2 4; do
  echo "== VS_DIRECT_TB=$tb" >> gpurun_out/r2q/direct_tb.txt
  VS_DIRECT_TB=$tb timeout 300 python tools/fwd_layer_times.py fwd dgrad --small 2>&1 | grep -v amdgpu.ids >> gpurun_out/r2q/direct_tb.txt
done
cat gpurun_out/r2q/pytest.log gpurun_out/r2q/direct_tb.txt
