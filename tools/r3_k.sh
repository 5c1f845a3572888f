#!/bin/bash
# odd batch sizes through the whole step (robustness, not perf)
mkdir -p gpurun_out/r3_k
for wl in sf_txenc_train feat_fwd; do
for n in 4 12 20; do
  timeout 600 python bench.py --workload $wl --clips-per-gpu $n --steps 5 --warmup 2 --no-cpu-baseline --no-roofline \
    > gpurun_out/r3_k/${wl}_$n.json 2> gpurun_out/r3_k/${wl}_$n.err
  echo "$wl $n exit $?"; grep -v amdgpu.ids gpurun_out/r3_k/${wl}_$n.err | tail -2 | cut -c1-300
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r3_k/${wl}_$n.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"])
except Exception as e:
    print("no line", e)
PY
done; done
