"""Apply on load (the b -> c edge of a bottleneck, training): bn_apply + conv_fwd (+ stats) and conv_wgrad on the stored
activation against conv_fwd_aol / conv_wgrad_aol on the producer's raw output, at the slow pathway's c units (8 clips),
graph replay timing.  usage: python tools/aol_time.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vidsitu_amd import ops

SHAPES = [  # name, clips, inner width, T, H, W, Cout, blocks per step
    ("slow res2 c", 8, 64, 8, 56, 56, 256, 3),
    ("slow res3 c", 8, 128, 8, 28, 28, 512, 4),
    ("slow res4 c", 8, 256, 8, 14, 14, 1024, 6),
    ("slow res5 c", 8, 512, 8, 7, 7, 2048, 3),
]


def graph_time(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


def main():
    dev = torch.device("cuda", 0)
    K1, S1, P0 = (1, 1, 1), (1, 1, 1), (0, 0, 0)
    tot = [0.0] * 5
    print(f"{'layer':12s} {'rows':>7s} {'K':>4s} {'N':>5s} |  apply   fwd  fwd_aol | wgrad wgrad_aol | per block: stored -> on load")
    for name, n, cin, t, h, w, cout, cnt in SHAPES:
        y = ops.new_act(n, cin, t, h, w, device=dev).normal_()
        dy = ops.new_act(n, cout, t, h, w, device=dev).normal_()
        wgt = torch.randn(cout, 1, 1, 1, cin, device=dev).to(torch.bfloat16).permute(0, 4, 1, 2, 3)
        sc, sh = torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev)
        act = ops.new_act(n, cin, t, h, w, device=dev)
        out = ops.new_act(n, cout, t, h, w, device=dev)
        dw = torch.empty((cout, 1, 1, 1, cin), dtype=torch.float32, device=dev).permute(0, 4, 1, 2, 3)
        if not ops.conv_aol_ok(y, cout):
            print(f"{name:12s} no apply-on-load plan")
            continue
        ops.bn_apply(y, sc, sh, None, True, out=act)
        t_apply = graph_time(lambda: ops.bn_apply(y, sc, sh, None, True, out=act))
        t_fwd = graph_time(lambda: ops.conv_fwd(act, wgt, K1, S1, P0, out=out, stats=True))
        t_fwd_aol = graph_time(lambda: ops.conv_fwd_aol(y, wgt, sc, sh, out=out, stats=True))
        t_wg = graph_time(lambda: ops.conv_wgrad(dy, act, K1, S1, P0, out=dw))
        t_wg_aol = graph_time(lambda: ops.conv_wgrad_aol(dy, y, sc, sh, out=dw))
        a, b = t_apply + t_fwd + t_wg, t_fwd_aol + t_wg_aol
        print(f"{name:12s} {n*t*h*w:7d} {cin:4d} {cout:5d} | {t_apply:6.1f} {t_fwd:5.1f} {t_fwd_aol:8.1f} | {t_wg:5.1f} {t_wg_aol:9.1f} |"
              f" {a:6.1f} -> {b:6.1f} us  x{cnt}")
        for i, v in enumerate((t_apply, t_fwd, t_fwd_aol, t_wg, t_wg_aol)):
            tot[i] += v * cnt
    print(f"sum x count: apply {tot[0]:.0f}, fwd {tot[1]:.0f} -> {tot[2]:.0f}, wgrad {tot[3]:.0f} -> {tot[4]:.0f} us; "
          f"stored {tot[0] + tot[1] + tot[3]:.0f} -> on load {tot[2] + tot[4]:.0f} us")


if __name__ == "__main__":
    main()
