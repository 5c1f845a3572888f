"""Where do the device-to-device copies (`__amd_rocclr_copyBuffer`) and torch-native fills / element-wise kernels of one
eager train step come from?  torch.profiler with python stacks, grouped by the innermost vidsitu_amd frame.
    python tools/find_copies.py  (through gpurun; writes gpurun_out/find_copies.txt)
"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

from vidsitu_amd import synth_data
from vidsitu_amd.extended_config import get_cfg
from vidsitu_amd.mdl_selector import get_mdl_loss_eval
from vidsitu_amd.optim import ArenaAdam, ParamArena
from vidsitu_amd.train_step import TrainStep

dev = torch.device("cuda:0")
cfg = get_cfg({"mdl.mdl_name": "sf_base_txenc", "tx_dec.encoder_layers": 6})
comm = synth_data.make_comm(cfg)
torch.manual_seed(0)
sel = get_mdl_loss_eval(cfg)
mdl = sel["mdl"](cfg=cfg, comm=comm).to(dev).train()
loss_fn = sel["loss"](cfg, comm)
batch = synth_data.synth_batch(cfg, comm, bs=2, n_ev=4, seed=1234, device=dev, dtype=torch.bfloat16)
arena = ParamArena(mdl)
opt = ArenaAdam(arena, lr=cfg.train.lr, betas=(0.9, 0.99))
ts = TrainStep(mdl, loss_fn, arena, opt, batch, world=1, grad_fill="learn")
for _ in range(3):
    ts.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    ts.step()
    torch.cuda.synchronize()

WANT = ("aten::copy_", "aten::fill_", "aten::zero_", "aten::mul", "aten::add", "aten::clone", "aten::contiguous",
        "aten::_to_copy", "aten::cat", "aten::index_select", "aten::sum", "aten::div")
groups = collections.Counter()
shapes = collections.defaultdict(collections.Counter)
for ev in prof.events():
    if ev.name not in WANT:
        continue
    if ev.cpu_parent is not None and ev.cpu_parent.name in WANT:
        continue  # only the outermost aten op
    frame = "?"
    for s in ev.stack or []:
        if "vidsitu_amd" in s or "bench.py" in s or "tools/" in s:
            frame = s.split("/root/repo/")[-1] if "/root/repo/" in s else s
            break
    groups[(ev.name, frame)] += 1
    shapes[(ev.name, frame)][str(ev.input_shapes)[:80]] += 1
os.makedirs("gpurun_out", exist_ok=True)
with open("gpurun_out/find_copies.txt", "w") as f:
    for (name, frame), n in groups.most_common(60):
        line = f"{n:5d}  {name:18s} {frame}   {dict(shapes[(name, frame)].most_common(3))}"
        print(line)
        f.write(line + "\n")
    kern = collections.Counter()
    for ev in prof.events():
        if ev.device_type == torch.autograd.DeviceType.CUDA and ("rocclr" in ev.name or "at::native" in ev.name or "Memcpy" in ev.name or "Memset" in ev.name):
            kern[ev.name[:100]] += 1
    for k, n in kern.most_common(20):
        line = f"{n:5d}  device: {k}"
        print(line)
        f.write(line + "\n")
