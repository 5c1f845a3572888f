"""Decode-step GEMMs of GPT-2 medium at M rows (default 50 = 10 sentences x beam 5): time per call,
weight-streaming rate and fp32 MFMA rate (eager launches: run under rocprofv3 --kernel-trace for
kernel times, tools/skinny_prof.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidsitu_amd import ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda:0")
shapes = [("c_attn", 3072, 1024), ("attn.c_proj", 1024, 1024), ("mlp.c_fc", 4096, 1024),
          ("mlp.c_proj", 1024, 4096), ("lm_head", 50259, 1024)]
for name, n, k in shapes:
    x = torch.randn(M, k, device=dev)
    # several weight copies so that consecutive calls do not hit a warm cache
    wts = [torch.randn(n, k, device=dev) / k ** 0.5 for _ in range(8 if n < 10000 else 3)]
    b = torch.randn(n, device=dev)
    for label in ("",):
        for _ in range(3):
            for wt in wts:
                ops.gemm_nt(x, wt, b)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            for wt in wts:
                ops.gemm_nt(x, wt, b)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (reps * len(wts))
        print(f"{name:12s} N={n:6d} K={k:5d} {label:9s}: {us:7.1f} us  {n*k*4/us/1e6:6.2f} TB/s weights  "
              f"{2*64*n*k/us/1e6:6.1f} TFLOP/s (64 padded rows)")
