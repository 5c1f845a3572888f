"""Microbenchmark of the implicit-GEMM conv kernel on the heavy SlowFast shapes (batch 8).
usage: python tools/conv_microbench.py [reps] [tile ids ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidsitu_amd import ops
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
tiles = [int(a) for a in sys.argv[2:]] or [None]
rings = [int(a) for a in os.environ.get("RINGS", "0").split(",")]
dgrad = os.environ.get("DGRAD") == "1"
# name, N, Cin, T, H, W, Cout, k, s, p
SHAPES = [
    ("s4.a  1024->256 [3,1,1]", 8, 1024, 8, 14, 14, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    ("s4.b  256->256  [1,3,3]", 8, 256, 8, 14, 14, 256, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("s4.c  256->1024 [1,1,1]", 8, 256, 8, 14, 14, 1024, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("s5.a  2048->512 [3,1,1]", 8, 2048, 8, 7, 7, 512, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    ("s5.b  512->512  [1,3,3]", 8, 512, 8, 7, 7, 512, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("s3.b  128->128  [1,3,3]", 8, 128, 8, 28, 28, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("s3.c  128->512  [1,1,1]", 8, 128, 8, 28, 28, 512, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("s2.b  64->64    [1,3,3]", 8, 64, 8, 56, 56, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("s2.c  64->256   [1,1,1]", 8, 64, 8, 56, 56, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("f.s2.c 8->32    [1,1,1]", 8, 8, 32, 56, 56, 32, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("f.s2.b 8->8     [1,3,3]", 8, 8, 32, 56, 56, 8, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
]
for name, n, cin, t, h, w, cout, k, s, p in SHAPES:
    x = ops.new_act(n, cin, t, h, w, dev); x.normal_()
    wt = (torch.randn(cout, *k, cin, device=dev) / (cin * k[0] * k[1] * k[2]) ** 0.5).to(ops.BF16).permute(0, 4, 1, 2, 3)
    ys = ops.conv_out_shape(x.shape, cout, k, s, p)
    flops = 2.0 * ys[0] * ys[2] * ys[3] * ys[4] * cout * cin * k[0] * k[1] * k[2]
    byts = 2.0 * (x.numel() + ys[0] * ys[1] * ys[2] * ys[3] * ys[4] + wt.numel())
    row = f"{name:26s}"
    for tile, ring in [(t, r) for t in tiles for r in rings]:
        fn = lambda: ops.conv_fwd(x, wt, k, s, p, stats=True, tile=tile, ring=ring)
        try:
            for _ in range(3): fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): fn()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            row += f" | t{tile}r{ring}: {ms*1e3:6.1f} us {flops/ms/1e9:7.1f} TF {byts/ms/1e6:6.0f} GB/s"
        except Exception as e:
            row += f" | t{tile}: ERR"
    print(row)
