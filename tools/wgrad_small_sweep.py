"""Weight-gradient plan sweep on the small-channel layers of the fast pathway (8..64 channels, 10^5..10^6 positions):
tile x block slots, each launch (+ slab reduce) alone on the GPU, graph replay.  usage: python tools/wgrad_small_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidsitu_amd import ops
from tools.wgrad_sweep import gt
dev = torch.device("cuda:0")
SH = [
    ("s2.p1.b 8->8 [1,3,3]", 8, 32, 56, 56, 8, (1, 3, 3), (0, 1, 1)),
    ("s2.p1.a 32->8 [3,1,1]", 32, 32, 56, 56, 8, (3, 1, 1), (1, 0, 0)),
    ("s2.p1.c 8->32 [1,1,1]", 8, 32, 56, 56, 32, (1, 1, 1), (0, 0, 0)),
    ("s3.p1.b 16->16 [1,3,3]", 16, 32, 28, 28, 16, (1, 3, 3), (0, 1, 1)),
    ("s3.p1.a 64->16 [3,1,1]", 64, 32, 28, 28, 16, (3, 1, 1), (1, 0, 0)),
    ("s3.p1.c 16->64 [1,1,1]", 16, 32, 28, 28, 64, (1, 1, 1), (0, 0, 0)),
    ("s4.p1.b 32->32 [1,3,3]", 32, 32, 14, 14, 32, (1, 3, 3), (0, 1, 1)),
    ("s4.p1.a 128->32 [3,1,1]", 128, 32, 14, 14, 32, (3, 1, 1), (1, 0, 0)),
]
# tiles: 4 = 32x128, 5 = 32x64, 6 = 16x128, 7 = 16x64
for name, cin, t, h, w, cout, k, p in SH:
    x = ops.new_act(8, cin, t, h, w, dev); x.normal_()
    dy = ops.new_act(8, cout, t, h, w, dev); dy.normal_()
    dw = torch.empty((cout, *k, cin), dtype=torch.float32, device=dev).permute(0, 4, 1, 2, 3)
    by = 2.0 * (x.numel() + dy.numel())
    row = f"{name:24s} ideal@5TB/s {by / 5e6:5.1f} plan {gt(lambda: ops.conv_wgrad(dy, x, k, (1, 1, 1), p, out=dw)):6.1f} |"
    for tile in (4, 5, 6, 7):
        for slots in (256, 384, 512, 768):
            try:
                us = gt(lambda: ops.conv_wgrad(dy, x, k, (1, 1, 1), p, out=dw, tile=tile, slots=slots))
                row += f" t{tile}s{slots}:{us:6.1f}"
            except Exception as e:
                row += f" t{tile}s{slots}:  ERR"
        row += " |"
    print(row, flush=True)
