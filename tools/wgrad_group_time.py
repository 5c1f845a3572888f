"""A ResBlock's weight gradients as separate launches (vs_conv_wgrad: kernel + slab reduce each) against ONE grouped launch
(vs_conv_wgrad_group), slow pathway of SlowFast-R50 at the bench shape, each form replayed 20x from a hipGraph alone on
the chip; + a numeric check of the grouped results against the separate ones.  usage: python tools/wgrad_group_time.py [--clips=N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidsitu_amd import ops

dev = torch.device("cuda:0")
N = int(next((a.split("=")[1] for a in sys.argv if a.startswith("--clips=")), 8))
REPS = 20


def graph_time(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REPS):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / REPS)
    return best * 1e3


def unit(cin, cout, k, s, t, hw):
    p = (k[0] // 2, k[1] // 2, k[2] // 2)
    x = ops.new_act(N, cin, t, hw, hw, dev); x.normal_()
    ys = ops.conv_out_shape(x.shape, cout, k, s, p)
    dy = ops.new_act(*ys, device=dev); dy.normal_()
    dw = torch.empty((cout, *k, cin), dtype=torch.float32, device=dev).permute(0, 4, 1, 2, 3)
    return dy, x, k, s, p, dw


# (name, [units]) : slow pathway blocks, T = 8
BLOCKS = [
    ("s3.res1 (a,b,c)", [unit(512, 128, (1, 1, 1), (1, 1, 1), 8, 28), unit(128, 128, (1, 3, 3), (1, 1, 1), 8, 28),
                         unit(128, 512, (1, 1, 1), (1, 1, 1), 8, 28)]),
    ("s4.res0 (sc,a,b,c)", [unit(640, 1024, (1, 1, 1), (1, 2, 2), 8, 28), unit(640, 256, (3, 1, 1), (1, 1, 1), 8, 28),
                            unit(256, 256, (1, 3, 3), (1, 2, 2), 8, 28), unit(256, 1024, (1, 1, 1), (1, 1, 1), 8, 14)]),
    ("s4.res1 (a,b,c)", [unit(1024, 256, (3, 1, 1), (1, 1, 1), 8, 14), unit(256, 256, (1, 3, 3), (1, 1, 1), 8, 14),
                         unit(256, 1024, (1, 1, 1), (1, 1, 1), 8, 14)]),
    ("s5.res0 (sc,a,b,c)", [unit(1280, 2048, (1, 1, 1), (1, 2, 2), 8, 14), unit(1280, 512, (3, 1, 1), (1, 1, 1), 8, 14),
                            unit(512, 512, (1, 3, 3), (1, 2, 2), 8, 14), unit(512, 2048, (1, 1, 1), (1, 1, 1), 8, 7)]),
    ("s5.res1 (a,b,c)", [unit(2048, 512, (3, 1, 1), (1, 1, 1), 8, 7), unit(512, 512, (1, 3, 3), (1, 1, 1), 8, 7),
                         unit(512, 2048, (1, 1, 1), (1, 1, 1), 8, 7)]),
    ("s2.res1 (a,b,c)", [unit(256, 64, (1, 1, 1), (1, 1, 1), 8, 56), unit(64, 64, (1, 3, 3), (1, 1, 1), 8, 56),
                         unit(64, 256, (1, 1, 1), (1, 1, 1), 8, 56)]),
]
print(f"{N} clips.  block: separate launches (sum of {'{'}kernel + reduce{'}'} chains) | grouped launch | worst rel. diff of dW")
tot = [0.0, 0.0]
for name, units in BLOCKS:
    if not ops.conv_wgrad_group_ok(units):
        print(f"{name:22s} not eligible"); continue
    def sep():
        for dy, x, k, s, p, dw in units:
            ops.conv_wgrad(dy, x, k, s, p, out=dw)
    def grp():
        ops.conv_wgrad_group(units)
    sep(); torch.cuda.synchronize()
    ref = [u[5].clone() for u in units]
    for u in units: u[5].zero_()
    grp(); torch.cuda.synchronize()
    worst = max(float((u[5] - r).abs().max() / r.abs().max()) for u, r in zip(units, ref))
    ts, tg = graph_time(sep), graph_time(grp)
    tot[0] += ts; tot[1] += tg
    print(f"{name:22s} {ts:8.1f} us | {tg:8.1f} us | {worst:.2e}", flush=True)
print(f"sum {tot[0]:.1f} -> {tot[1]:.1f} us")
