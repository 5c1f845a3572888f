"""Evaluation: conv b -> conv c of a fast-pathway bottleneck as one launch (vs_conv_fwd_bc) against the two vs_conv_fwd
launches it replaces, at the bench shapes (8 clips), graph replay timing.  usage: python tools/bc_fuse_time.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vidsitu_amd import ops

SHAPES = [  # name, clips, Cb, T, H, W, Cc, blocks per step
    ("fast res2 b->c", 8, 8, 32, 56, 56, 32, 3),
    ("fast res3 b->c", 8, 16, 32, 28, 28, 64, 4),
    ("fast res4 b->c", 8, 32, 32, 14, 14, 128, 6),
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
    k, s, p = (1, 3, 3), (1, 1, 1), (0, 1, 1)
    tot = [0.0, 0.0]
    for name, n, cb, t, h, w, cc, cnt in SHAPES:
        x = ops.new_act(n, cb, t, h, w, device=dev).normal_()
        res = ops.new_act(n, cc, t, h, w, device=dev).normal_()
        wb = torch.randn(cb, 1, 3, 3, cb, device=dev).to(torch.bfloat16).permute(0, 4, 1, 2, 3)
        wc = torch.randn(cc, 1, 1, 1, cb, device=dev).to(torch.bfloat16).permute(0, 4, 1, 2, 3)
        sb, hb = torch.rand(cb, device=dev) + 0.5, torch.randn(cb, device=dev)
        sc, hc = torch.rand(cc, device=dev) + 0.5, torch.randn(cc, device=dev)
        mid = ops.new_act(n, cb, t, h, w, device=dev)
        out = ops.new_act(n, cc, t, h, w, device=dev)

        def two():
            ops.conv_fwd(x, wb, k, s, p, out=mid, scale=sb, shift=hb, relu=True)
            ops.conv_fwd(mid, wc, (1, 1, 1), (1, 1, 1), (0, 0, 0), out=out, scale=sc, shift=hc, residual=res, relu=True)

        def one():
            ops.conv_fwd_bc(x, wb, k, s, p, sb, hb, wc, sc, hc, residual=res, relu=True, out=out)

        if not ops.conv_fwd_bc_fusable(x, wb, k, s, p, cc):
            print(f"{name:16s} not fusable")
            continue
        t2, t1 = graph_time(two), graph_time(one)
        byts = 2.0 * n * t * h * w * (cb + 2 * cc)
        print(f"{name:16s} rows {n*t*h*w:7d} {cb:3d}->{cb:3d}->{cc:4d} | two launches {t2:6.1f} us | one {t1:6.1f} us "
              f"({byts / t1 / 1e6:5.2f} TB/s of its own bytes) | x{cnt}")
        tot[0] += t2 * cnt
        tot[1] += t1 * cnt
    print(f"sum x count: two launches {tot[0]:.1f} us, one launch {tot[1]:.1f} us")


if __name__ == "__main__":
    main()
