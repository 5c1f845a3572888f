"""Which BN units of the 8-clip train step still run a BN-backward REDUCE pass of their own (vs_bn_bwd_reduce: one read
of dz and y) instead of getting their sums from the data gradient that produced their dz?  One eager step, every
ops.bn_bwd call logged with the unit's name."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidsitu_amd import ops, synth_data, trunk
from vidsitu_amd.extended_config import get_cfg
from vidsitu_amd.mdl_selector import get_mdl_loss_eval
from vidsitu_amd.optim import ArenaAdam, ParamArena

dev = torch.device("cuda:0")
cfg = get_cfg({"mdl.mdl_name": "sf_base"})
comm = synth_data.make_comm(cfg)
torch.manual_seed(0)
sel = get_mdl_loss_eval(cfg)
mdl = sel["mdl"](cfg=cfg, comm=comm).to(dev).train()
names = {id(m): n for n, m in mdl.named_modules()}
batch = synth_data.synth_batch(cfg, comm, bs=2, n_ev=4, seed=1234, device=dev, dtype=torch.bfloat16)
arena = ParamArena(mdl); opt = ArenaAdam(arena, lr=1e-4)
log = []
orig_bwd = trunk._Unit.bwd.__func__ if hasattr(trunk._Unit.bwd, "__func__") else trunk._Unit.bwd
orig_bn_bwd = ops.bn_bwd
cur = {}
def bn_bwd(dz, z, y, *a, **k):
    fused = k.get("partial") is not None
    log.append((cur.get("name"), tuple(y.shape), fused, "pool" if k.get("pool_src") is not None else ""))
    return orig_bn_bwd(dz, z, y, *a, **k)
ops.bn_bwd = bn_bwd
def bwd(rec, dz, *a, **k):
    cur["name"] = names.get(id(rec["bn"]), "?")
    return orig_bwd(rec, dz, *a, **k)
trunk._Unit.bwd = staticmethod(bwd)
opt.zero_grad()
loss = sel["loss"](cfg, comm)(mdl(batch), batch)["loss"]
loss.backward()
torch.cuda.synchronize()
tot = 0
for name, shp, fused, pool in log:
    n, c, t, h, w = shp
    mb = n * c * t * h * w * 2 / 1e6
    if not fused:
        tot += 2 * mb
    print(f"{name:45s} {str(shp):28s} {'sums from the dgrad' if fused else 'REDUCE PASS'} {pool} {mb:7.1f} MB")
print(f"{sum(1 for l in log if not l[2])} reduce passes of {len(log)} units, {tot / 1e3:.2f} GB read by them")
