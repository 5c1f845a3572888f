"""Apply kernels that finalize for themselves (vs_bn_apply_fin / vs_bn_bwd_apply_fin) against finalize + apply as two
launches, per SlowFast-R50 layer shape at 8 clips: a hipGraph of 20 dependent repetitions each (one stream), us per
unit.  usage: python tools/bn_fin_time.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vidsitu_amd import ops

# (name, rows, channels, partial rows fwd, residual)
SHAPES = [
    ("s4.p0.b", 12544, 256, 196, False), ("s4.p0.c", 12544, 1024, 98, True), ("s4.p0.sc", 12544, 1024, 98, False),
    ("s5.p0.a", 12544, 512, 98, False), ("s5.p0.b", 3136, 512, 49, False), ("s5.p0.c", 3136, 2048, 25, True),
    ("s5.p1.b", 12544, 64, 196, False), ("s5.p1.c", 12544, 256, 196, True),
    ("s3.p0.c", 50176, 512, 784, True), ("s2.p0.a", 200704, 64, 3136, False), ("s2.p0.c", 200704, 256, 3136, True),
    ("s2.p1.a", 802816, 8, 1568, False), ("s4.p1.a", 200704, 32, 392, False),
]


class BN:
    def __init__(self, c, dev):
        self.weight, self.bias = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.1
        self.running_mean, self.running_var = torch.zeros(c, device=dev), torch.ones(c, device=dev)
        self.momentum, self.eps = 0.1, 1e-5


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
    print(f"{'layer':10s} {'rows':>7s} {'C':>5s} {'parts':>5s} | fwd: fin+apply  fused | bwd: fin+apply  fused   (us per unit)")
    for name, rows, c, nparts, res in SHAPES:
        sets = max(2, min(8, int(600e6 // (rows * c * 2 * 3))))
        mk = lambda: [ops.new_act(1, c, 1, 1, rows, dev).normal_() for _ in range(sets)]
        ys, rs, zs, dzs = mk(), mk(), mk(), mk()
        bn = BN(c, dev)
        partials = torch.randn(nparts, 2, c, device=dev).abs_() + 1.0
        partials[:, 1] += partials[:, 0] ** 2
        k = [0]

        def fwd(fused):
            i = k[0] % sets
            k[0] += 1
            if fused:
                ops.bn_apply_fin(partials, rows, bn, ys[i], rs[i] if res else None, True, out=zs[i], want_bits=False)
            else:
                sc, sh, _, _ = ops.bn_finalize(partials, rows, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                               0.1, 1e-5, train=True)
                ops.bn_apply(ys[i], sc, sh, rs[i] if res else None, True, out=zs[i])
        tf = [graph_time(lambda: fwd(f)) for f in (False, True)] if ops.bn_fin_fusable(nparts, c) else [float("nan")] * 2
        mean, invstd = torch.randn(c, device=dev) * 0.1, torch.rand(c, device=dev) + 0.5
        dg, db = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
        nb = min(nparts, 256)
        part_b = torch.randn(nb, 2, c, device=dev)

        def bwd(fused):
            i = k[0] % sets
            k[0] += 1
            ops.BN_FIN_FUSE = fused
            ops.bn_bwd(dzs[i], None, ys[i], mean, invstd, bn.weight, True, False, dy_out=zs[i], dgamma=dg, dbeta=db,
                       beta=bn.bias, partial=part_b)
        tb = [graph_time(lambda: bwd(f)) for f in (False, True)]
        ops.BN_FIN_FUSE = True
        print(f"{name:10s} {rows:7d} {c:5d} {nparts:5d} |    {tf[0]:8.1f} {tf[1]:8.1f}  |    {tb[0]:8.1f} {tb[1]:8.1f}")


if __name__ == "__main__":
    main()
