"""Stem kernels at the bench shape (8 clips, 224 x 224): forward (train epilogue) and weight gradient, graph replay."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidsitu_amd import ops
dev = torch.device("cuda:0")
REPS = 10

def gt(fn):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REPS): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / REPS)
    return best * 1e3

for name, cout, kt, t in (("slow stem 3->64 [1,7,7]", 64, 1, 8), ("fast stem 3->8 [5,7,7]", 8, 5, 32)):
    x = torch.randn(8, 3, t, 224, 224, device=dev)
    x4 = ops.pack_input(x, 4)
    wt = torch.randn(cout, 3, kt, 7, 7, device=dev) / (147 * kt) ** 0.5
    wp = ops.pack_stem_weight(wt)
    y, part = ops.stem_conv_fwd(x4, wp, cout, kt, stats=True)
    fwd = gt(lambda: ops.stem_conv_fwd(x4, wp, cout, kt, stats=True))
    dy = torch.randn_like(y.float()).to(ops.BF16) if False else y
    wg = gt(lambda: ops.stem_conv_wgrad(y, x4, kt))
    print(f"{name}: fwd {fwd:6.1f} us   wgrad {wg:6.1f} us", flush=True)
