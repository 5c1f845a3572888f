"""Debug aid for the in-launch split-K plan: one layer, the split launch against torch, where the differences are."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import torch
import torch.nn.functional as F
from vidsitu_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
n, cin, t, h, w, cout, k, s, p = 2, 256, 4, 14, 14, 256, (1, 3, 3), (1, 1, 1), (0, 1, 1)
x = torch.randn(n, cin, t, h, w, generator=g).bfloat16().float()
wgt = (torch.randn(cout, cin, *k, generator=g) / (cin * 9) ** 0.5).bfloat16().float()
ref = F.conv3d(x, wgt, stride=s, padding=p)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from gpu_utils import to_act, to_w
xa, wa = to_act(x, dev), to_w(wgt, dev)
for il in ("naive", "forced64x128r3", "forced128r2", None, False, True):
    if il == "naive":
        y, part = ops.conv_fwd(xa, wa, k, s, p, naive=True)
    elif il == "forced64x128r3":
        y, part = ops.conv_fwd(xa, wa, k, s, p, stats=True, tile=1, ring=3)
    elif il == "forced128r2":
        y, part = ops.conv_fwd(xa, wa, k, s, p, stats=True, tile=0, ring=2)
    else:
        y, part = ops.conv_fwd(xa, wa, k, s, p, stats=True, halo=False, pw=False, deep=False, splitk_il=il)
    print("y", y.float().cpu()[0, :6, 0, 0, 0].tolist(), "ref", ref[0, :6, 0, 0, 0].tolist())
    torch.cuda.synchronize()
    yf = y.float().cpu()
    bad = ~torch.isfinite(yf)
    err = (yf - ref).abs()
    err[bad] = 0
    print("il", il, "nan", int(bad.sum()), "of", yf.numel(), "max err", float(err.max()), "ref max", float(ref.abs().max()))
    if bad.any():
        idx = bad.nonzero()
        print("  first bad", idx[:5].tolist(), "bad per clip", bad.flatten(1).sum(1).tolist())
    big = (err > 0.05).nonzero()
    print("  errors > 0.05:", big.shape[0], big[:5].tolist())
ws = ops._workspace(1, dev, "splitk")
print("ws bytes", ws.numel(), "counters", ws[:256].view(torch.int32).tolist()[:32])
d = ops.make_desc(tuple(x.shape), cin, tuple(ref.shape), cout, k, s, p, (1 << 29) | (1 << 21) | (1 << 23) | (1 << 27))
out = (C.c_int * 5)(); ops._lib.load().vs_conv_plan(C.byref(d), 0, out); print("plan", list(out), "ws need", ops._lib.load().vs_conv_workspace_bytes(C.byref(d), 0))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for il in (False, True):
    for _ in range(3):
        ops.conv_fwd(xa, wa, k, s, p, stats=True, halo=False, pw=False, deep=False, splitk_il=il)
    e0.record()
    for _ in range(20):
        ops.conv_fwd(xa, wa, k, s, p, stats=True, halo=False, pw=False, deep=False, splitk_il=il)
    e1.record(); torch.cuda.synchronize()
    print("il", il, "us per launch (eager)", e0.elapsed_time(e1) / 20 * 1e3)
