"""Microbenchmark of vs_conv_wgrad on the heavy SlowFast shapes (batch 8); hipGraph replay so that
launch latency is not measured.  usage: RINGS=1,2,3 python tools/wgrad_microbench.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidsitu_amd import ops
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rings = [int(a) for a in os.environ.get("RINGS", "1,2,3").split(",")]
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
    ("f.s3.b 16->16   [1,3,3]", 8, 16, 32, 28, 28, 16, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("f.s4.a 128->32  [3,1,1]", 8, 128, 32, 14, 14, 32, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
]
for name, n, cin, t, h, w, cout, k, s, p in SHAPES:
    x = ops.new_act(n, cin, t, h, w, dev); x.normal_()
    ys = ops.conv_out_shape(x.shape, cout, k, s, p)
    dy = ops.new_act(*ys, device=dev); dy.normal_()
    flops = 2.0 * ys[0] * ys[2] * ys[3] * ys[4] * cout * cin * k[0] * k[1] * k[2]
    row = f"{name:26s}"
    for ring in rings:
        out = ops.conv_wgrad(dy, x, k, s, p, ring=ring)
        fn = lambda: ops.conv_wgrad(dy, x, k, s, p, out=out, ring=ring)
        for _ in range(2): fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(reps): fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        row += f" | r{ring}: {ms*1e3:6.1f} us {flops/ms/1e9:6.1f} TF"
    print(row)
