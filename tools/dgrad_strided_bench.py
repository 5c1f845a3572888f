"""Strided-conv dgrad (MODE 2) with / without stride-class tiling, hipGraph replay."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidsitu_amd import ops
dev = torch.device("cuda:0")
SHAPES = [  # name, N, Cin, T, H, W (input), Cout, k, s, p   -- the strided convs of SlowFast-R50 at batch 8
    ("s3.b0 128 3x3 s2", 8, 128, 8, 56, 56, 128, (1, 3, 3), (1, 2, 2), (0, 1, 1)),
    ("s4.b0 256 3x3 s2", 8, 256, 8, 28, 28, 256, (1, 3, 3), (1, 2, 2), (0, 1, 1)),
    ("s5.b0 512 3x3 s2", 8, 512, 8, 14, 14, 512, (1, 3, 3), (1, 2, 2), (0, 1, 1)),
    ("s3.sc 320->512 1x1 s2", 8, 320, 8, 56, 56, 512, (1, 1, 1), (1, 2, 2), (0, 0, 0)),
    ("s4.sc 640->1024 1x1 s2", 8, 640, 8, 28, 28, 1024, (1, 1, 1), (1, 2, 2), (0, 0, 0)),
    ("s5.sc 1280->2048 1x1 s2", 8, 1280, 8, 14, 14, 2048, (1, 1, 1), (1, 2, 2), (0, 0, 0)),
    ("fuse 32->64 [7,1,1] s4", 8, 32, 32, 28, 28, 64, (7, 1, 1), (4, 1, 1), (3, 0, 0)),
]
for name, n, cin, t, h, w, cout, k, s, p in SHAPES:
    xs = (n, cin, t, h, w)
    ys = ops.conv_out_shape(xs, cout, k, s, p)
    dy = ops.new_act(*ys, device=dev); dy.normal_()
    wt = torch.randn((cin, *k, cout), device=dev).to(ops.BF16).permute(0, 4, 1, 2, 3)
    row = f"{name:26s}"
    for noclass in (True, False):
        fn = lambda: ops.conv_dgrad(dy, wt, xs, k, s, p, noclass=noclass)
        for _ in range(2): fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20): fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        row += f" | {'plain ' if noclass else 'class '} {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us"
    print(row)
