"""vs_linear_bwd_fused vs the three separate launches, repeated under GPU contention (N copies of this process share
cuda:0): is the horizontally fused launch itself irreproducible?  usage: python tools/probes/linear_fused_race.py launch COPIES ITERS"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if sys.argv[1] == "launch":
    import io
    from vidsitu_amd import dist_launch
    out, err = io.StringIO(), io.StringIO()
    os.environ["VS_ITERS"] = sys.argv[3]
    rc = dist_launch.launch_ranks(int(sys.argv[2]), [sys.executable, os.path.abspath(__file__), "child"], out=out, err=err,
                                  check_devices=False)
    print("\n".join(ln for ln in err.getvalue().splitlines() if "RACE" in ln or "Error" in ln))
    sys.exit(rc)
import torch
from vidsitu_amd import ops
rank = int(os.environ.get("RANK", "0"))
dev = torch.device("cuda", 0)
g = torch.Generator(device="cpu").manual_seed(5)
iters = int(os.environ.get("VS_ITERS", "500"))
for (m, n, k, relu) in ((4, 1024, 1024, False), (4, 1024, 2304, True), (4, 3072, 1024, False), (4, 1024, 1024, True), (8, 1024, 1024, False)):
    dy = torch.randn(m, n, generator=g).to(dev)
    x = torch.randn(m, k, generator=g).to(dev)
    w = torch.randn(n, k, generator=g).to(dev)
    wt = w.t().contiguous()
    y = torch.randn(m, n, generator=g).to(dev) if relu else None
    ops._LINEAR_BWD_FUSED = False
    dx0, dw0, db0 = ops.linear_bwd(dy, x, w, wt=wt, relu_y=y)
    torch.cuda.synchronize()
    ops._LINEAR_BWD_FUSED = True
    bad = [0, 0, 0]
    keep = []
    for it in range(iters):
        # fresh outputs every time, a few other allocations in between (as the autograd pass has)
        junk = torch.empty(m * k + 17 * (it % 5), device=dev).normal_()
        dx, dw, db = ops.linear_bwd(dy, x, w, wt=wt, relu_y=y)
        keep = [junk]
        if it % 1 == 0:
            bad[0] += int(not torch.equal(dx, dx0))
            bad[1] += int(not torch.equal(dw, dw0))
            bad[2] += int(not torch.equal(db, db0))
    sys.stderr.write(f"RACE rank {rank} M{m} N{n} K{k} relu={relu}: dx/dw/db mismatches {bad} of {iters}\n")
