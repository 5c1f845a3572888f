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
if "--fast" in sys.argv:  # fast-pathway blocks (T = 32, 1/8 of the channels)
    BLOCKS = [
        ("fast s5.res1", [unit(256, 64, (3, 1, 1), (1, 1, 1), 32, 7), unit(64, 64, (1, 3, 3), (1, 1, 1), 32, 7),
                          unit(64, 256, (1, 1, 1), (1, 1, 1), 32, 7)]),
        ("fast s4.res1", [unit(128, 32, (3, 1, 1), (1, 1, 1), 32, 14), unit(32, 32, (1, 3, 3), (1, 1, 1), 32, 14),
                          unit(32, 128, (1, 1, 1), (1, 1, 1), 32, 14)]),
        ("fast s3.res1", [unit(64, 16, (3, 1, 1), (1, 1, 1), 32, 28), unit(16, 16, (1, 3, 3), (1, 1, 1), 32, 28),
                          unit(16, 64, (1, 1, 1), (1, 1, 1), 32, 28)]),
        ("fast s5.res0", [unit(128, 256, (1, 1, 1), (1, 2, 2), 32, 14), unit(128, 64, (3, 1, 1), (1, 1, 1), 32, 14),
                          unit(64, 64, (1, 3, 3), (1, 2, 2), 32, 14), unit(64, 256, (1, 1, 1), (1, 1, 1), 32, 7)]),
    ]


def rep(units, n):  # the same block n times (independent tensors)
    out = []
    for _ in range(n):
        out += [unit(x.shape[1], dy.shape[1], k, s_, x.shape[2], x.shape[3]) for dy, x, k, s_, p, dw in units]
    return out


if "--span" in sys.argv:  # two consecutive blocks of a stage as one group (<= 8 items)
    BLOCKS = [("2 x s3.res1", rep(BLOCKS[0][1], 2)), ("2 x s4.res1", rep(BLOCKS[2][1], 2)), ("2 x s5.res1", rep(BLOCKS[4][1], 2)),
              ("s4.res0 + s4.res1", BLOCKS[1][1] + rep(BLOCKS[2][1], 1)), ("s5.res0 + s5.res1", BLOCKS[3][1] + rep(BLOCKS[4][1], 1))]
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
