"""Rank 0 repeats the isolated fused few-row linear backward (and its separate-launch form) and compares bitwise;
rank 1, ANOTHER PROCESS on the same GPU, runs something else in a loop: 'conv' = MFMA convolution kernels (LDS-DMA ring),
'bn' = BN element-wise streaming passes, 'linear' = the same linear kernels.  Which neighbour corrupts the linear kernel?
usage: python tools/probes/linear_vs_mfma_neighbor.py launch NEIGHBOUR ITERS"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if sys.argv[1] == "launch":
    import io
    from vidsitu_amd import dist_launch
    out, err = io.StringIO(), io.StringIO()
    os.environ["VS_NEIGH"], os.environ["VS_ITERS"] = sys.argv[2], sys.argv[3]
    rc = dist_launch.launch_ranks(2, [sys.executable, os.path.abspath(__file__), "child"], out=out, err=err, check_devices=False)
    print("\n".join(ln[:300] for ln in err.getvalue().splitlines() if "NEIGH" in ln or "Error" in ln)[:6000])
    sys.exit(rc)
import torch
from vidsitu_amd import ops
rank = int(os.environ.get("RANK", "0"))
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
iters = int(os.environ["VS_ITERS"])
flag = "/tmp/vs_neigh_done"
if rank == 1:
    kind = os.environ["VS_NEIGH"]
    if kind.startswith("same_"):
        sys.exit(0)
    if os.path.exists(flag):
        os.remove(flag)
    if kind == "conv":
        x = ops.new_act(8, 256, 8, 28, 28, dev).normal_()
        w = torch.randn(256, 3, 3, 256, device=dev).to(ops.BF16).view(256, 1, 3, 3, 256).permute(0, 4, 1, 2, 3)
        fn = lambda: ops.conv_fwd(x, w, (1, 3, 3), (1, 1, 1), (0, 1, 1), halo=False)
    elif kind == "conv_reg":  # the same convolution on the register-staged pipeline: MFMA + LDS, no LDS-DMA
        x = ops.new_act(8, 256, 8, 28, 28, dev).normal_()
        w = torch.randn(256, 3, 3, 256, device=dev).to(ops.BF16).view(256, 1, 3, 3, 256).permute(0, 4, 1, 2, 3)
        fn = lambda: ops.conv_fwd(x, w, (1, 3, 3), (1, 1, 1), (0, 1, 1), ring=1, halo=False)
    elif kind == "conv_halo":
        x = ops.new_act(8, 256, 8, 28, 28, dev).normal_()
        w = torch.randn(256, 3, 3, 256, device=dev).to(ops.BF16).view(256, 1, 3, 3, 256).permute(0, 4, 1, 2, 3)
        fn = lambda: ops.conv_fwd(x, w, (1, 3, 3), (1, 1, 1), (0, 1, 1), halo="force")
    elif kind == "bn":
        y = ops.new_act(8, 256, 8, 56, 56, dev).normal_()
        sc, sh = torch.ones(256, device=dev), torch.zeros(256, device=dev)
        fn = lambda: ops.bn_apply(y, sc, sh, None, True)
    else:
        dy, xx, ww = torch.randn(8, 1024, device=dev), torch.randn(8, 1024, device=dev), torch.randn(1024, 1024, device=dev)
        wt = ww.t().contiguous()
        fn = lambda: ops.linear_bwd(dy, xx, ww, wt=wt)
    n = 0
    while not os.path.exists(flag) and n < 400000:
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        n += 50
    sys.stderr.write(f"NEIGH rank 1 ({kind}) ran {n} launches\n")
    sys.exit(0)
same = os.environ["VS_NEIGH"].startswith("same_")
if same:  # the neighbour is a second STREAM of this process: 3000 conv launches enqueued ahead on it per round
    xs = ops.new_act(8, 256, 8, 28, 28, dev).normal_()
    ws = torch.randn(256, 3, 3, 256, device=dev).to(ops.BF16).view(256, 1, 3, 3, 256).permute(0, 4, 1, 2, 3)
    side = torch.cuda.Stream()
    def feed(nl=1500):
        with torch.cuda.stream(side):
            for _ in range(nl):
                ops.conv_fwd(xs, ws, (1, 3, 3), (1, 1, 1), (0, 1, 1), halo=False)
else:
    time.sleep(3.0)  # let the neighbour start
g = torch.Generator(device="cpu").manual_seed(5)
for fused in (True, False):
    for (m, n, k) in ((8, 1024, 1024), (4, 1024, 2304)):
        dy = torch.randn(m, n, generator=g).to(dev)
        x = torch.randn(m, k, generator=g).to(dev)
        w = torch.randn(n, k, generator=g).to(dev)
        wt = w.t().contiguous()
        ops._LINEAR_BWD_FUSED = fused
        dx0, dw0, db0 = ops.linear_bwd(dy, x, w, wt=wt)
        torch.cuda.synchronize()
        bad = [0, 0, 0]
        for it in range(iters):
            if same and it % 300 == 0:
                feed()
            dx, dw, db = ops.linear_bwd(dy, x, w, wt=wt)
            bad[0] += int(not torch.equal(dx, dx0)); bad[1] += int(not torch.equal(dw, dw0)); bad[2] += int(not torch.equal(db, db0))
            if fused and not torch.equal(dw, dw0) and bad[1] <= 2:
                ref = (dy.double().t() @ x.double())  # exact
                ne = (dw.double() - ref).abs() > 1e-3 * ref.abs().max()
                ne0 = (dw0.double() - ref).abs() > 1e-3 * ref.abs().max()
                rows_, cols_ = ne.any(1).nonzero().flatten(), ne.any(0).nonzero().flatten()
                sys.stderr.write(f"NEIGHD it {it}: wrong vs fp64 in this launch {int(ne.sum())} (in the reference launch {int(ne0.sum())}); rows "
                                 f"{rows_[:8].tolist()} .. {rows_[-4:].tolist()} ({rows_.numel()}), cols {cols_[:8].tolist()} .. {cols_[-4:].tolist()} ({cols_.numel()})\n")
                idx = ne.nonzero()[:3]
                for (r, c) in idx.tolist():
                    terms = (dy[:, r].double() * x[:, c].double()).tolist()
                    sys.stderr.write(f"NEIGHD   dw[{r}][{c}] got {float(dw[r, c]):.6f} want {float(ref[r, c]):.6f}; terms {['%.4f' % t for t in terms]}\n")
        sys.stderr.write(f"NEIGH rank 0 fused={fused} M{m} N{n} K{k}: dx/dw/db mismatches {bad} of {iters} (neighbour {os.environ['VS_NEIGH']})\n")
open(flag, "w").write("1")
