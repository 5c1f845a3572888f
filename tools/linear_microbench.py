"""Small-M fp32 linear (TxEncoder / heads / GPT-2 decode shapes), hipGraph replay."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidsitu_amd import ops
dev = torch.device("cuda:0")
for m, n, k in [(8, 1024, 1024), (8, 1024, 2304), (8, 1564, 1024), (40, 1024, 1024), (50, 1024, 1024), (50, 4096, 1024), (50, 1024, 4096), (50, 50259, 1024), (600, 3072, 1024), (600, 1024, 4096)]:
    x = torch.randn(m, k, device=dev); ws = [torch.randn(n, k, device=dev) for _ in range(8)]
    b = torch.randn(n, device=dev)
    y = torch.empty(m, n, device=dev)
    fn = lambda i: ops.gemm_nt(x, ws[i % 8], b, out=y)
    for i in range(8): fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(40): fn(i)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 40 * 1e3
    print(f"M={m:4d} N={n:6d} K={k:5d}: {us:7.1f} us  {n*k*4/us/1e3:7.1f} GB/s weights  {2.0*m*n*k/us/1e6:7.2f} TFLOP/s")
