"""Per-call HIP-event timing of every conv launch (fwd / dgrad / wgrad) in one training step,
grouped by GEMM shape.  Run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidsitu_amd import ops, synth_data
from vidsitu_amd.extended_config import get_cfg
from vidsitu_amd.mdl_selector import get_mdl_loss_eval

dev = torch.device("cuda:0")
cfg = get_cfg({"mdl.mdl_name": "sf_base"})
comm = synth_data.make_comm(cfg)
torch.manual_seed(0)
sel = get_mdl_loss_eval(cfg)
mdl = sel["mdl"](cfg=cfg, comm=comm).to(dev).train()
batch = synth_data.synth_batch(cfg, comm, bs=2, n_ev=4, device=dev, dtype=torch.bfloat16)
loss_fn = sel["loss"](cfg, comm)
rec = []
names = {"conv_fwd": ops.conv_fwd, "conv_dgrad": ops.conv_dgrad, "conv_wgrad": ops.conv_wgrad,
         "stem_conv_fwd": ops.stem_conv_fwd, "bn_bwd": ops.bn_bwd, "bn_apply": ops.bn_apply,
         "bn_finalize": ops.bn_finalize}
def wrap(name, fn):
    def inner(*a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = fn(*a, **kw); e1.record()
        t = [x for x in a if torch.is_tensor(x)]
        key = name + " " + " ".join("x".join(map(str, x.shape)) for x in t[:2])
        if name in ("conv_fwd", "conv_dgrad", "conv_wgrad"):
            key += f" k{a[2] if name!='conv_dgrad' else a[3]} s{a[3] if name!='conv_dgrad' else a[4]}"
        rec.append((key, e0, e1)); return out
    return inner
for it in range(3):
    if it == 2:
        for n, f in names.items(): setattr(ops, n, wrap(n, f))
    loss = loss_fn(mdl(batch), batch)["loss"]; loss.backward()
torch.cuda.synchronize()
agg = {}
for key, e0, e1 in rec:
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1)
tot = {}
for k, (n, ms) in agg.items():
    tot[k.split()[0]] = tot.get(k.split()[0], 0) + ms
print({k: round(v, 3) for k, v in tot.items()}, "total", round(sum(tot.values()), 3))
for k, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{ms:8.3f} ms  x{n:2d}  {k}")
