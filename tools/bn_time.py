"""Achieved HBM rate of the BN element-wise passes at the train step's shapes (8 clips, SlowFast-R50):
forward apply (+ residual + ReLU + mask bits for the c units) and backward apply, graph replay timing.
Sweep knobs (read once by the library): VS_BN_TARGET (blocks a launch aims for), VS_BN_NBMAX (row batches
per block).  usage: python tools/bn_time.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vidsitu_amd import ops

SHAPES = [  # (name, clips x T x H x W positions, channels, count per step)
    ("s2.p0.c", 200704, 256, 3), ("s3.p0.c", 50176, 512, 4), ("s4.p0.c", 12544, 1024, 6), ("s5.p0.c", 3136, 2048, 3),
    ("s2.p0.ab", 200704, 64, 6), ("s3.p0.ab", 50176, 128, 8), ("s4.p0.ab", 12544, 256, 12), ("s5.p0.ab", 3136, 512, 6),
    ("s2.p1.c", 802816, 32, 3), ("s3.p1.c", 200704, 64, 4), ("s4.p1.c", 50176, 128, 6),
    ("s2.p1.ab", 802816, 8, 6), ("s3.p1.ab", 200704, 16, 8), ("s4.p1.ab", 50176, 32, 12),
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
    print(f"VS_BN_TARGET={os.environ.get('VS_BN_TARGET', '-')} VS_BN_NBMAX={os.environ.get('VS_BN_NBMAX', '-')}")
    tot_f = tot_b = 0.0
    for name, rows, c, cnt in SHAPES:
        n = 8
        t = rows // n
        # several buffer sets so that a replay does not hit the Infinity Cache
        sets = max(2, min(8, int(600e6 // (rows * c * 2 * 3))))
        ys = [ops.new_act(n, c, t, 1, 1, dev) for _ in range(sets)]
        rs = [ops.new_act(n, c, t, 1, 1, dev) for _ in range(sets)]
        zs = [ops.new_act(n, c, t, 1, 1, dev) for _ in range(sets)]
        for a in ys + rs:
            a.normal_()
        sc, sh = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)
        is_c = name.endswith(".c")
        k = [0]
        bits = []

        def fwd():
            i = k[0] % sets
            k[0] += 1
            if is_c:
                bits[:] = [ops.bn_apply(ys[i], sc, sh, rs[i], True, out=zs[i], want_bits=True)[1]]
            else:
                ops.bn_apply(ys[i], sc, sh, None, True, out=zs[i])
        tf = graph_time(fwd)
        bytes_f = rows * c * 2 * (3 if is_c else 2) + (rows * c // 8 if is_c else 0)
        mean, invstd, gamma, beta = torch.randn(c, device=dev), torch.rand(c, device=dev) + 0.5, sc, sh
        dg, db = torch.randn(c, device=dev), torch.randn(c, device=dev)
        part = torch.zeros((1, 2, c), device=dev)
        zb = bits[0] if is_c else None

        def bwd():
            i = k[0] % sets
            k[0] += 1
            # partial given: the reduce pass is skipped (the dgrad epilogue emitted the sums); finalize + apply run
            ops.bn_bwd(rs[i], None, ys[i], mean, invstd, gamma, True, False, dy_out=zs[i], dgamma=dg, dbeta=db,
                       beta=beta, zbits=zb, partial=part)
        tb = graph_time(bwd)
        bytes_b = rows * c * 2 * 3 + (rows * c // 8 if is_c else 0)
        tot_f += tf * cnt
        tot_b += tb * cnt
        print(f"{name:10s} rows {rows:7d} C {c:5d} | fwd apply {tf:7.1f} us {bytes_f / tf / 1e6:6.2f} TB/s | "
              f"bwd finalize+apply {tb:7.1f} us {bytes_b / tb / 1e6:6.2f} TB/s")
    print(f"sum x count: fwd {tot_f / 1e3:.3f} ms, bwd {tot_b / 1e3:.3f} ms")


if __name__ == "__main__":
    main()
