"""How many (dgrad, wgrad) pairs one train step of the bench model issues as single launches, and which conv launches
of the backward pass stay separate (entry point, kernel label)."""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidsitu_amd import ops, synth_data
from vidsitu_amd.extended_config import get_cfg
from vidsitu_amd.mdl_selector import get_mdl_loss_eval
from vidsitu_amd.optim import ArenaAdam, ParamArena
from vidsitu_amd.train_step import TrainStep

dev = torch.device("cuda", 0)
cfg = get_cfg({"mdl.mdl_name": "sf_base_txenc", "tx_dec.encoder_layers": 6})
comm = synth_data.make_comm(cfg)
torch.manual_seed(0)
sel = get_mdl_loss_eval(cfg)
mdl = sel["mdl"](cfg=cfg, comm=comm).to(dev).train()
batch = synth_data.synth_batch(cfg, comm, bs=2, n_ev=4, seed=1234, device=dev, dtype=torch.bfloat16)
arena = ParamArena(mdl)
ts = TrainStep(mdl, sel["loss"](cfg, comm), arena, ArenaAdam(arena, lr=1e-4), batch, world=1, use_dist=False)
ts.step(); torch.cuda.synchronize()
n0 = ops.conv_pair_count()
import ctypes as C
from vidsitu_amd import _lib
lib = _lib.load()
seen = collections.Counter()
orig = _lib.call


def probe(name, *a):
    if name in ("vs_conv_dgrad", "vs_conv_dgrad_ex", "vs_conv_dgrad_bnstats", "vs_conv_wgrad"):
        d = a[3]._obj
        out = (C.c_int * 5)()
        if name == "vs_conv_wgrad":  # (vs_conv_plan describes the forward / dgrad kernels only)
            seen[("wgrad", (), (d.kT, d.kH, d.kW), d.Cin, d.Cout)] += 1
        else:
            lib.vs_conv_plan(C.byref(d), 1, out)
            seen[(name.replace("vs_conv_", ""), tuple(out), (d.kT, d.kH, d.kW), d.Cin, d.Cout)] += 1
    return orig(name, *a)
_lib.call = probe
ops._lib.call = probe
ts.step(); torch.cuda.synchronize()
print("pairs issued in one step:", ops.conv_pair_count() - n0)
for k, c in sorted(seen.items(), key=lambda kv: (-kv[1])):
    print(c, k)
