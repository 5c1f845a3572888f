"""A/B of tile x ring on the pointwise (1x1x1) layers of the slow pathway, train epilogue (stats) and eval epilogue
(affine + residual + relu); graph replay.  usage: python tools/pw_ab.py"""
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

K1, S1, P0 = (1, 1, 1), (1, 1, 1), (0, 0, 0)
# name, Cin, T, H, W, Cout, stride
SH = [("s2.c", 64, 8, 56, 56, 256, S1), ("s3.c", 128, 8, 28, 28, 512, S1), ("s4.c", 256, 8, 14, 14, 1024, S1),
      ("s5.c", 512, 8, 7, 7, 2048, S1), ("s2.sc", 80, 8, 56, 56, 256, S1), ("s3.sc", 320, 8, 56, 56, 512, (1, 2, 2)),
      ("s4.sc", 640, 8, 28, 28, 1024, (1, 2, 2)), ("s2.a", 256, 8, 56, 56, 64, S1), ("s3.a", 512, 8, 28, 28, 128, S1)]
print("tiles: 0=128x128 1=64x128 2=128x64 3=64x64 6=256x128 7=128x256 ; rN = LDS-DMA ring stages (r1 = register staged)")
for name, cin, t, h, w, cout, s in SH:
    x = ops.new_act(8, cin, t, h, w, dev); x.normal_()
    wt = (torch.randn(cout, 1, 1, 1, cin, device=dev) / cin ** 0.5).to(ops.BF16).permute(0, 4, 1, 2, 3)
    ys = ops.conv_out_shape(x.shape, cout, K1, s, P0)
    out = ops.new_act(*ys, device=dev)
    res = ops.new_act(*ys, device=dev); res.normal_()
    sc = torch.rand(cout, device=dev) + 0.5; sh = torch.randn(cout, device=dev)
    by = 2.0 * (x.numel() / (s[1] * s[2]) + out.numel() + cout * cin)
    for label, kw in (("train", dict(stats=True)), ("eval", dict(scale=sc, shift=sh, residual=res, relu=True))):
        row = f"{name:6s} {label:5s} ideal@5TB/s {by / 5e6 + (out.numel() * 2 / 5e6 if label == 'eval' else 0):5.1f} |"
        base = gt(lambda: ops.conv_fwd(x, wt, K1, s, P0, out=out, **kw))
        row += f" pw {base:5.1f} |"
        base = gt(lambda: ops.conv_fwd(x, wt, K1, s, P0, out=out, pw=False, **kw))
        row += f" igemm auto {base:5.1f} |"
        for tile in ((0, 1, 2, 3, 6, 7) if "--tiles" in sys.argv else ()):
            for ring in (1, 2, 3):
                try:
                    us = gt(lambda: ops.conv_fwd(x, wt, K1, s, P0, out=out, tile=tile, ring=ring, **kw))
                    row += f" t{tile}r{ring} {us:5.1f}"
                except Exception as e:
                    row += f" t{tile}r{ring}  ERR "
            row += " |"
        print(row, flush=True)
