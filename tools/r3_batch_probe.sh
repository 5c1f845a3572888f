#!/bin/bash
# batch-size probe: how the same kernels fill the chip at 16 / 32 clips per GPU (not the headline config)
mkdir -p gpurun_out/r3_i
for wl in sf_txenc_train feat_fwd; do
for n in 8 16 32; do
  timeout 600 python bench.py --workload $wl --clips-per-gpu $n --steps 20 --warmup 5 --no-cpu-baseline --no-roofline \
    > gpurun_out/r3_i/${wl}_$n.json 2> gpurun_out/r3_i/${wl}_$n.err
  echo "$wl $n exit $?"; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r3_i/${wl}_$n.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["config"].get("frac_of_bf16_mfma_peak"))
except Exception as e:
    print("no line", e)
PY
done; done
