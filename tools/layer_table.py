"""Analytic table of every conv GEMM of SlowFast-R50 at the bench shape (8 clips): M, N, K, GFLOP,
algorithmic MB, FLOP/B, number of 128x128 tiles, and the time at a given rate.  CPU only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vidsitu_amd.extended_config import get_cfg
from vidsitu_amd.trunk import VideoTrunk, ResBlock

def rows(n=8, hw=224, t_fast=32):
    cfg = get_cfg({})
    tr = VideoTrunk(cfg.sf_mdl)
    out = []
    def add(name, conv, shape):
        n_, c, t, h, w = shape
        k, s, p = conv.k, conv.s, conv.p
        to, ho, wo = (t + 2*p[0]-k[0])//s[0]+1, (h+2*p[1]-k[1])//s[1]+1, (w+2*p[2]-k[2])//s[2]+1
        M, N, K = n_*to*ho*wo, conv.cout, conv.cin*k[0]*k[1]*k[2]
        out.append((name, M, N, K, k, s, n_*t*h*w*conv.cin))
        return (n_, conv.cout, to, ho, wo)
    shp = [(n, 3, t_fast//4, hw, hw), (n, 3, t_fast, hw, hw)]
    cur = []
    for p in range(2):
        st = getattr(tr.s1, f"pathway{p}_stem")
        o = add(f"s1.p{p}.stem", st.conv, shp[p])
        cur.append((o[0], o[1], o[2], (o[3]+2-3)//2+1, (o[4]+2-3)//2+1))
    def fuse(name, f, cur):
        o = add(name, f.conv_f2s, cur[1])
        cur[0] = (cur[0][0], cur[0][1] + o[1], cur[0][2], cur[0][3], cur[0][4])
    fuse("s1_fuse", tr.s1_fuse, cur)
    for k in range(2, 6):
        st = getattr(tr, f"s{k}")
        for p in range(2):
            x = cur[p]
            for i, blk in enumerate(st.blocks(p)):
                b2 = blk.branch2
                pre = f"s{k}.p{p}.b{i}"
                if blk.has_sc:
                    add(pre + ".sc", blk.branch1, x)
                a = add(pre + ".a", b2.a, x)
                b = add(pre + ".b", b2.b, a)
                x = add(pre + ".c", b2.c, b)
            cur[p] = x
        if k < 5:
            fuse(f"s{k}_fuse", getattr(tr, f"s{k}_fuse"), cur)
    return out

if __name__ == "__main__":
    rate = float(sys.argv[1]) if len(sys.argv) > 1 else 1000.0  # TFLOP/s
    tot = 0; agg = {}
    print(f"{'layer':14s} {'M':>8s} {'N':>5s} {'K':>5s} {'GFLOP':>8s} {'MB':>7s} {'F/B':>6s} {'t128':>5s} {'us@rate':>8s}")
    for name, M, N, K, k, s, xin in rows():
        fl = 2.0*M*N*K
        by = 2.0*(xin + M*N + N*K)
        t128 = ((M+127)//128)*((N+127)//128)
        tot += fl
        key = (M, N, K, k, s)
        a = agg.setdefault(key, [0, name]); a[0] += 1
        print(f"{name:14s} {M:8d} {N:5d} {K:5d} {fl/1e9:8.2f} {by/1e6:7.1f} {fl/by:6.0f} {t128:5d} {fl/rate/1e6:8.1f}")
    print("total GFLOP", tot/1e9, "distinct shapes", len(agg))
    print("\ndistinct shapes by total GFLOP:")
    for (M, N, K, k, s), (cnt, name) in sorted(agg.items(), key=lambda kv: -2.0*kv[0][0]*kv[0][1]*kv[0][2]*kv[1][0]):
        fl = 2.0*M*N*K
        print(f"  x{cnt:2d} {name:14s} M{M:7d} N{N:5d} K{K:5d} k{k} s{s}  {fl*cnt/1e9:8.1f} GFLOP total, {fl/1e9:6.2f} each, tiles128 {((M+127)//128)*((N+127)//128)}")
