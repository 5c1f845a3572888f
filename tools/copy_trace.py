"""Where do the D2D copies of one eager training step come from?  torch.profiler, grouped by the Python
frame that issued them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from vidsitu_amd import synth_data
from vidsitu_amd.extended_config import get_cfg
from vidsitu_amd.mdl_selector import get_mdl_loss_eval
from vidsitu_amd.optim import ArenaAdam, ParamArena

dev = torch.device("cuda:0")
cfg = get_cfg({"mdl.mdl_name": "sf_base_txenc"})
comm = synth_data.make_comm(cfg)
sel = get_mdl_loss_eval(cfg)
mdl = sel["mdl"](cfg=cfg, comm=comm).to(dev).train()
arena = ParamArena(mdl)
opt = ArenaAdam(arena, lr=1e-4)
batch = synth_data.synth_batch(cfg, comm, bs=2, n_ev=4, device=dev, dtype=torch.bfloat16)
loss_fn = sel["loss"](cfg, comm)


def step():
    opt.zero_grad()
    loss = loss_fn(mdl(batch), batch)["loss"]
    loss.backward()
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
ev = [e for e in prof.events() if e.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::to", "aten::_to_copy", "aten::fill_", "aten::zero_", "aten::add_", "aten::add", "aten::mul")]
from collections import Counter
c = Counter()
for e in ev:
    st = [s for s in (e.stack or []) if "vidsitu_amd" in s or "bench" in s]
    c[(e.name, st[0] if st else "?")] += 1
for (n, s), k in c.most_common(40):
    print(k, n, s)
