"""Weight gradients whose output tile has < 128 rows (Cout <= 64): register-staged pipeline (ring 1) vs LDS-DMA ring
2 / 3, each launch (+ slab reduce) alone on the GPU, graph replay.  usage: python tools/wgrad_ring_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidsitu_amd import ops
from tools.pool_time import gt
dev = torch.device("cuda:0")
SH = [
    ("s2.b 64->64 [1,3,3]", 64, 8, 56, 56, 64, (1, 3, 3), (0, 1, 1)),
    ("s2.a 256->64 [1,1,1]", 256, 8, 56, 56, 64, (1, 1, 1), (0, 0, 0)),
    ("s2.p1.b 8->8 [1,3,3]", 8, 32, 56, 56, 8, (1, 3, 3), (0, 1, 1)),
    ("s2.p1.a 32->8 [3,1,1]", 32, 32, 56, 56, 8, (3, 1, 1), (1, 0, 0)),
    ("s3.p1.b 16->16 [1,3,3]", 16, 32, 28, 28, 16, (1, 3, 3), (0, 1, 1)),
    ("s3.p1.a 64->16 [3,1,1]", 64, 32, 28, 28, 16, (3, 1, 1), (1, 0, 0)),
    ("s4.p1.b 32->32 [1,3,3]", 32, 32, 14, 14, 32, (1, 3, 3), (0, 1, 1)),
    ("s4.p1.a 128->32 [3,1,1]", 128, 32, 14, 14, 32, (3, 1, 1), (1, 0, 0)),
    ("s5.p1.b 64->64 [1,3,3]", 64, 32, 7, 7, 64, (1, 3, 3), (0, 1, 1)),
    ("s5.p1.a 256->64 [3,1,1]", 256, 32, 7, 7, 64, (3, 1, 1), (1, 0, 0)),
]
for name, cin, t, h, w, cout, k, p in SH:
    x = ops.new_act(8, cin, t, h, w, dev); x.normal_()
    dy = ops.new_act(8, cout, t, h, w, dev); dy.normal_()
    dw = torch.empty((cout, *k, cin), dtype=torch.float32, device=dev).permute(0, 4, 1, 2, 3)
    row = f"{name:26s}"
    for ring in (0, 1, 2, 3):
        try:
            row += f" ring{ring}: {gt(lambda: ops.conv_wgrad(dy, x, k, (1, 1, 1), p, out=dw, ring=ring)):6.1f}"
        except Exception as e:
            row += f" ring{ring}:   ERR"
    print(row, flush=True)
