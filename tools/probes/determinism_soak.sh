#!/bin/bash
# two independent 1000-step trainings of the full model through main_dist.py (hipGraph replays, two pathway streams,
# pair launches, merged reduces): same seed, same synthetic batches -> the saved parameters and Adam moments must be
# bit for bit the same
mkdir -p gpurun_out/soak
for r in a b; do
  python main_dist.py soak_$r --steps=1000 --misc.tmp_path=gpurun_out/soak > gpurun_out/soak/run_$r.log 2>&1
  echo "run $r exit $?"; grep -E "steps|saved|valid" gpurun_out/soak/run_$r.log | cut -c1-220
done
python - <<'PY'
import torch
a = torch.load("gpurun_out/soak/models/soak_a.pth", map_location="cpu", weights_only=False)
b = torch.load("gpurun_out/soak/models/soak_b.pth", map_location="cpu", weights_only=False)
def walk(x, y, path=""):
    bad = []
    if torch.is_tensor(x):
        if not torch.equal(x, y):
            bad.append(path)
    elif isinstance(x, dict):
        for k in x:
            bad += walk(x[k], y[k], f"{path}/{k}")
    elif isinstance(x, (list, tuple)):
        for i, (u, v) in enumerate(zip(x, y)):
            bad += walk(u, v, f"{path}/{i}")
    return bad
bad = walk(a, b)
n = sum(1 for _ in a["model_state_dict"]) if "model_state_dict" in a else -1
print("tensors in the model state:", n, "| differing entries:", len(bad), bad[:5])
PY
rm -rf gpurun_out/soak/models
