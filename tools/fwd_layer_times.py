"""Per-layer time of every distinct conv GEMM of SlowFast-R50 at the bench shape (8 clips), each launch
alone on the GPU replayed from a hipGraph: forward (eval epilogue: folded BN + ReLU), dgrad, wgrad.
Prints us, TFLOP/s, algorithmic GB/s and the fraction of the binding roof (max of flops / 2.5 PF and
bytes / 8 TB/s).  usage: python tools/fwd_layer_times.py [fwd|dgrad|wgrad ...] [--clips=N] [--only=s4.,s5.]
(--clips: the same layers at another batch, e.g. 32 = each kernel's steady state; --only: name substrings)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidsitu_amd import ops
from tools.layer_table import rows

dev = torch.device("cuda:0")
kinds = [a for a in sys.argv[1:] if not a.startswith("--")] or ["fwd"]
HALO = "force" if "--halo" in sys.argv else (False if "--nohalo" in sys.argv else True)
DEEP = "force" if "--deep" in sys.argv else (False if "--nodeep" in sys.argv else True)  # conv_deep.hip: forced / off / plan
REPS = 20
NCLIPS = int(next((a.split("=")[1] for a in sys.argv if a.startswith("--clips=")), 8))
ONLY = next((a.split("=")[1].split(",") for a in sys.argv if a.startswith("--only=")), None)


def graph_time(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REPS):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / REPS)
    return best * 1e3  # us


agg = {}
for name, M, N, K, k, s, xin in rows(n=NCLIPS):
    key = (M, N, K, k, s)
    a = agg.setdefault(key, [0, name, xin]); a[0] += 1
tot = {kd: [0.0, 0.0] for kd in kinds}
print(f"{'layer':14s} {'x':>2s} {'M':>7s} {'N':>5s} {'K':>5s} " + " ".join(f"{kd + ' us':>9s} {'TF/s':>6s} {'GB/s':>6s} {'roof':>5s}" for kd in kinds))
for (M, N, K, k, s), (cnt, name, xin) in sorted(agg.items(), key=lambda kv: -2.0 * kv[0][0] * kv[0][1] * kv[0][2] * kv[1][0]):
    if "stem" in name:
        continue
    if ONLY and not any(o in name for o in ONLY):
        continue
    if "--small" in sys.argv and not (N <= 32 and K <= 192):  # the register-resident small-channel kernel's layers
        continue
    taps = k[0] * k[1] * k[2]
    cin = K // taps
    # reconstruct the input shape: rows() gives M of the output; invert the strides
    # (all strided layers here: spatial stride 2 with pad k//2, temporal stride 4 for the fuse convs)
    p = (k[0] // 2, k[1] // 2, k[2] // 2)
    n = NCLIPS
    pos_in = xin // cin
    # find (t, h, w) of the input from the known stage geometry
    cands = [(t, hw, hw) for t in (8, 32) for hw in (56, 28, 14, 7) if n * t * hw * hw == pos_in]
    t, h, w = cands[0]
    x = ops.new_act(n, cin, t, h, w, dev); x.normal_()
    wt = (torch.randn(N, *k, cin, device=dev) / K ** 0.5).to(ops.BF16).permute(0, 4, 1, 2, 3)
    ys = ops.conv_out_shape(x.shape, N, k, s, p)
    assert ys[0] * ys[2] * ys[3] * ys[4] == M, (name, ys, M)
    fl = 2.0 * M * N * K
    by = 2.0 * (xin + M * N + N * K)
    sc = torch.rand(N, device=dev) + 0.5
    sh = torch.randn(N, device=dev)
    row = f"{name:14s} {cnt:2d} {M:7d} {N:5d} {K:5d} "
    for kd in kinds:
        if kd == "fwd":
            out = ops.new_act(*ys, device=dev)
            fn = lambda: ops.conv_fwd(x, wt, k, s, p, out=out, scale=sc, shift=sh, relu=True, halo=HALO, deep=DEEP)
        elif kd == "dgrad":
            dy = ops.new_act(*ys, device=dev); dy.normal_()
            wtt = ops.weight_transpose(wt)
            dx = ops.new_act(*x.shape, device=dev)
            fn = lambda: ops.conv_dgrad(dy, wtt, tuple(x.shape), k, s, p, out=dx, halo=HALO, deep=DEEP)
        else:
            dy = ops.new_act(*ys, device=dev); dy.normal_()
            dw = torch.empty((N, *k, cin), dtype=torch.float32, device=dev).permute(0, 4, 1, 2, 3)
            fn = lambda: ops.conv_wgrad(dy, x, k, s, p, out=dw, deep=DEEP)
        us = graph_time(fn)
        roof = max(fl / 2.5e15, by / 8e12) * 1e6 / us
        tot[kd][0] += us * cnt
        tot[kd][1] += fl * cnt
        row += f"{us:9.1f} {fl / us / 1e6:6.0f} {by / us / 1e3:6.0f} {roof:5.2f} "
    print(row, flush=True)
for kd in kinds:
    print(f"{kd}: sum over layers (x count) {tot[kd][0] / 1e3:.3f} ms, {tot[kd][1] / tot[kd][0] / 1e6:.0f} TFLOP/s (stems excluded)")
