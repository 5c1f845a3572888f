"""Outputs of the small-channel (direct) conv kernel on fast-pathway shapes, saved for a cross-process comparison
(VS_DIRECT_TB is read once per process).  usage: VS_DIRECT_TB=n python tools/direct_dump.py out.pt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidsitu_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(5)
out = {}
# name, N, Cin, T, H, W, Cout, k, s, p
SH = [("s2a", 2, 32, 32, 16, 16, 8, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
      ("s2a0", 2, 8, 32, 16, 16, 8, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
      ("s2b", 2, 8, 32, 16, 16, 8, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
      ("s2c", 2, 8, 32, 16, 16, 32, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
      ("s3b0", 2, 16, 32, 16, 16, 16, (1, 3, 3), (1, 2, 2), (0, 1, 1)),
      ("s3b", 2, 16, 32, 8, 8, 16, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
      ("s3sc", 2, 32, 32, 16, 16, 64, (1, 1, 1), (1, 2, 2), (0, 0, 0)),
      ("fuse1", 2, 8, 32, 16, 16, 16, (5, 1, 1), (4, 1, 1), (2, 0, 0)),
      ("odd", 3, 8, 5, 13, 11, 8, (1, 3, 3), (1, 1, 1), (0, 1, 1))]
for name, n, cin, t, h, w, cout, k, s, p in SH:
    x = ops.new_act(n, cin, t, h, w, dev); x.copy_(torch.randn(x.shape, generator=g).to(dev))
    wt = (torch.randn(cout, *k, cin, generator=g) / (cin * k[0] * k[1] * k[2]) ** 0.5).to(dev).to(ops.BF16).permute(0, 4, 1, 2, 3)
    ys = ops.conv_out_shape(x.shape, cout, k, s, p)
    y, part = ops.conv_fwd(x, wt, k, s, p, stats=True)
    out[name + ".y"] = y.float().cpu(); out[name + ".part"] = part.sum(0).cpu()
    sc = torch.rand(cout, generator=g).to(dev) + 0.5; sh = torch.randn(cout, generator=g).to(dev)
    r = ops.new_act(*ys, device=dev); r.copy_(torch.randn(r.shape, generator=g).to(dev))
    y2, _ = ops.conv_fwd(x, wt, k, s, p, scale=sc, shift=sh, residual=r, relu=True)
    out[name + ".y2"] = y2.float().cpu()
    dy = ops.new_act(*ys, device=dev); dy.copy_(torch.randn(dy.shape, generator=g).to(dev))
    wtt = ops.weight_transpose(wt)
    out[name + ".dx"] = ops.conv_dgrad(dy, wtt, tuple(x.shape), k, s, p).float().cpu()
    rr = ops.new_act(*x.shape, device=dev); rr.copy_(torch.randn(rr.shape, generator=g).to(dev))
    bits = torch.randint(0, 256, (ops.act_rows(rr), cin // 8), generator=g, dtype=torch.uint8).to(dev)
    out[name + ".dxm"] = ops.conv_dgrad(dy, wtt, tuple(x.shape), k, s, p, residual=rr, residual_bits=bits).float().cpu()
    acc = rr.clone()
    ops.conv_dgrad(dy, wtt, tuple(x.shape), k, s, p, residual=acc, inplace=True)
    out[name + ".dxi"] = acc.float().cpu()
    import ctypes as C
    d = ops.make_desc(tuple(x.shape), cin, ys, cout, k, s, p, 0)
    pl = (C.c_int * 5)(); ops._lib.load().vs_conv_plan(C.byref(d), 0, pl)
    pd = (C.c_int * 5)(); ops._lib.load().vs_conv_plan(C.byref(d), 1, pd)
    out[name + ".plan"] = torch.tensor(list(pl) + list(pd))
torch.save(out, sys.argv[1])
