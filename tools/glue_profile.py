"""Which Python lines of the train step launch torch-native kernels (copies, fills, elementwise)?"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from vidsitu_amd import synth_data
from vidsitu_amd.extended_config import get_cfg
from vidsitu_amd.mdl_selector import get_mdl_loss_eval
from vidsitu_amd.optim import ArenaAdam, ParamArena
dev = torch.device("cuda:0")
cfg = get_cfg({"mdl.mdl_name": "sf_base_txenc", "tx_dec.encoder_layers": 6})
comm = synth_data.make_comm(cfg)
torch.manual_seed(0)
sel = get_mdl_loss_eval(cfg)
mdl = sel["mdl"](cfg=cfg, comm=comm).to(dev).train()
loss_fn = sel["loss"](cfg, comm)
batch = synth_data.synth_batch(cfg, comm, bs=2, n_ev=4, device=dev, dtype=torch.bfloat16)
arena = ParamArena(mdl); opt = ArenaAdam(arena, lr=1e-4)
def step():
    opt.zero_grad(); out = mdl(batch); loss = loss_fn(out, batch)["loss"]; loss.backward(); opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=False) as prof:
    step()
torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name.startswith("aten::") and ev.name in ("aten::copy_", "aten::fill_", "aten::mul", "aten::add_", "aten::add", "aten::zero_", "aten::clone", "aten::contiguous", "aten::_to_copy", "aten::gt", "aten::native_dropout", "aten::sum", "aten::div", "aten::mul_", "aten::zeros", "aten::ones", "aten::empty_strided"):
        st = [s for s in (ev.stack or []) if "vidsitu_amd" in s or "bench" in s or "glue_profile" in s]
        cnt[(ev.name, st[0] if st else "?")] += 1
for (name, where), n in cnt.most_common(40):
    print(f"{n:4d} {name:22s} {where}")
