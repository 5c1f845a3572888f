"""Pools / input packing at the bench shape, graph replay.  usage: python tools/pool_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidsitu_amd import ops
REPS = 20


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


dev = torch.device("cuda:0")
def main():
    for name, c, t in (("slow", 64, 8), ("fast", 8, 32)):
        x = ops.new_act(8, c, t, 112, 112, dev); x.normal_()
        y, idx = ops.maxpool_hw(x, want_idx=True)
        f = gt(lambda: ops.maxpool_hw(x, out=y, want_idx=True))
        dy = ops.new_act(*y.shape, device=dev); dy.normal_()
        b = gt(lambda: ops.maxpool_hw_bwd(dy, idx, tuple(x.shape)))
        xin = torch.randn(8, 3, t, 224, 224, device=dev).to(torch.bfloat16)
        pk = gt(lambda: ops.pack_input(xin, 4))
        print(f"{name}: maxpool fwd {f:6.1f} us  bwd {b:6.1f} us   pack_input(bf16 -> c4) {pk:6.1f} us", flush=True)


if __name__ == "__main__":
    main()
