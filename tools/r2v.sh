#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r2v; rm -f gpurun_out/r2v/*
for tb in 1 2 4; do VS_DIRECT_TB=$tb python tools/direct_dump.py /tmp/tb$tb.pt 2>&1 | grep -v amdgpu.ids; done
python - <<'PY' | tee gpurun_out/r2v/cmp.txt
import torch
a = torch.load("/tmp/tb1.pt")
for tb in (2, 4):
    b = torch.load(f"/tmp/tb{tb}.pt")
    for k in a:
        if k.endswith(".plan"):
            if tb == 2: print(k, a[k].tolist())
            continue
        d = (a[k] - b[k]).abs().max().item()
        if d > 0: print(f"TB={tb} {k}: max diff {d:.4e} (max {a[k].abs().max().item():.3e}), first bad idx {torch.nonzero((a[k]-b[k]).abs()>0)[0].tolist()}, n bad {(a[k]!=b[k]).sum().item()} of {a[k].numel()}")
print("done")
PY
