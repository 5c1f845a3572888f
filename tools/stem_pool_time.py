"""BN + ReLU + max-pool of the stems, fused vs separate launches (graph replay timing, both stem shapes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidsitu_amd import ops
from tools.bn_time import graph_time

dev = torch.device("cuda", 0)
for name, (n, c, t, h, w) in (("slow stem", (8, 64, 8, 112, 112)), ("fast stem", (8, 8, 32, 112, 112))):
    sets = 4
    ys = [ops.new_act(n, c, t, h, w, dev).normal_() for _ in range(sets)]
    sc, sh = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.3
    mean, invstd = torch.randn(c, device=dev) * 0.1, torch.rand(c, device=dev) + 0.5
    p0, i0 = ops.bn_apply_maxpool(ys[0], sc, sh)
    dps = [torch.randn_like(p0.float()).to(p0.dtype) for _ in range(sets)]
    dg, db = torch.empty(c, device=dev), torch.empty(c, device=dev)
    k = [0]

    def nxt():
        k[0] += 1
        return k[0] % sets
    t_sep = graph_time(lambda: ops.maxpool_hw(ops.bn_apply(ys[nxt()], sc, sh, None, True), want_idx=True))
    t_fus = graph_time(lambda: ops.bn_apply_maxpool(ys[nxt()], sc, sh))

    def bwd_sep():
        i = nxt()
        dz = ops.maxpool_hw_bwd(dps[i], i0, tuple(ys[i].shape))
        ops.bn_bwd(dz, None, ys[i], mean, invstd, sc, True, False, dgamma=dg, dbeta=db, beta=sh)

    def bwd_fus():
        i = nxt()
        ops.bn_bwd(None, None, ys[i], mean, invstd, sc, True, False, dgamma=dg, dbeta=db, beta=sh, pool_src=(dps[i], i0))
    b_sep, b_fus = graph_time(bwd_sep), graph_time(bwd_fus)
    print(f"{name}: forward apply+pool {t_sep:6.1f} us, fused {t_fus:6.1f} us | backward pool_bwd+reduce+finalize+apply "
          f"{b_sep:6.1f} us, fused {b_fus:6.1f} us")
