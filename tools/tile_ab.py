"""A/B of forced tile configs on a few dgrad / fwd shapes (graph replay).  usage: python tools/tile_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidsitu_amd import ops
dev = torch.device("cuda:0")
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

# name, kind, N, Cin, T, H, W, Cout, k, s, p
SH = [
    ("s2_fuse dgrad", "d", 8, 32, 32, 56, 56, 64, (7, 1, 1), (4, 1, 1), (3, 0, 0)),
    ("s3_fuse dgrad", "d", 8, 64, 32, 28, 28, 128, (7, 1, 1), (4, 1, 1), (3, 0, 0)),
    ("s1_fuse dgrad", "d", 8, 8, 32, 56, 56, 16, (7, 1, 1), (4, 1, 1), (3, 0, 0)),
    ("s4.p1.b fwd", "f", 8, 32, 32, 14, 14, 32, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("s4.p1.a fwd", "f", 8, 128, 32, 14, 14, 32, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    ("s4.p1.a dgrad", "d", 8, 128, 32, 14, 14, 32, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    ("s3.b0.sc dgrad", "d", 8, 320, 8, 56, 56, 512, (1, 1, 1), (1, 2, 2), (0, 0, 0)),
    ("s3.b0.b dgrad", "d", 8, 128, 8, 56, 56, 128, (1, 3, 3), (1, 2, 2), (0, 1, 1)),
]
for name, kind, n, cin, t, h, w, cout, k, s, p in SH:
    x = ops.new_act(n, cin, t, h, w, dev); x.normal_()
    wt = (torch.randn(cout, *k, cin, device=dev) / (cin * k[0] * k[1] * k[2]) ** 0.5).to(ops.BF16).permute(0, 4, 1, 2, 3)
    ys = ops.conv_out_shape(x.shape, cout, k, s, p)
    dy = ops.new_act(*ys, device=dev); dy.normal_()
    wtt = ops.weight_transpose(wt)
    row = f"{name:18s}"
    for tile in (None, 0, 1, 2, 3, 4):
        try:
            if kind == "f":
                out = ops.new_act(*ys, device=dev)
                fn = lambda: ops.conv_fwd(x, wt, k, s, p, out=out, stats=True, tile=tile)
            else:
                dx = ops.new_act(*x.shape, device=dev)
                fn = lambda: ops.conv_dgrad(dy, wtt, tuple(x.shape), k, s, p, out=dx, tile=tile)
            row += f" | t{tile}: {gt(fn):6.1f}"
        except Exception as e:
            row += f" | t{tile}: ERR"
    if kind == "d" and any(v != 1 for v in s):
        acc = ops.new_act(*x.shape, device=dev); acc.normal_()
        row += f" | inplace: {gt(lambda: ops.conv_dgrad(dy, wtt, tuple(x.shape), k, s, p, residual=acc, inplace=True)):6.1f}"
        res = ops.new_act(*x.shape, device=dev); res.normal_()
        dx = ops.new_act(*x.shape, device=dev)
        row += f" | +res: {gt(lambda: ops.conv_dgrad(dy, wtt, tuple(x.shape), k, s, p, residual=res, out=dx)):6.1f}"
    print(row, flush=True)
