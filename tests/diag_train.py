"""Per-unit train-mode diagnostics: HIP trunk vs oracle (checker script, run by hand on the GPU box:
python tests/diag_train.py; lives under tests/ because it imports the oracle)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
from oracle.slowfast_ref import VideoTrunk as RefTrunk, default_sf_cfg, randomize_bn, slow_index, ResBlock
from vidsitu_amd.trunk import VideoTrunk, _Unit

def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-20))

def run(arch, depth, width, frames, n, hw):
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    cfg = default_sf_cfg(arch, depth, width, frames)
    ref = RefTrunk(cfg); randomize_bn(ref, 3)
    ours = VideoTrunk(cfg); ours.load_state_dict(ref.state_dict()); ours.to(dev)
    g = torch.Generator().manual_seed(4)
    fast = torch.randn(n, 3, frames, hw, hw, generator=g)
    xs = [fast.index_select(2, slow_index(frames, 4)), fast] if arch == "slowfast" else [fast]
    ref.train(); ours.train()
    # oracle: conv outputs (pre-BN) and unit outputs, in execution order
    conv_out, names = {}, {}
    for name, m in ref.named_modules():
        if isinstance(m, nn.Conv3d):
            m.register_forward_hook(lambda mod, i, o, name=name: conv_out.__setitem__(name, o.detach().clone()))
    blk_out = {}
    for name, m in ref.named_modules():
        if isinstance(m, ResBlock):
            m.register_forward_hook(lambda mod, i, o, name=name: blk_out.__setitem__(name, o.detach().clone()))
    fr = ref.forward_features(xs)
    _Unit.trace = []
    fo = ours.forward_features([x.to(dev) for x in xs])
    conv_names = {id(m): n_ for n_, m in ours.named_modules()}
    print(f"=== {arch} {depth} n={n} hw={hw}")
    for conv, y, z, mean, invstd in _Unit.trace:
        nm = conv_names[id(conv)]
        yr = conv_out[nm]
        line = f"{nm:45s} conv-out rel {rel(y, yr):.3e}  mean-err {float((mean - yr.mean(dim=(0,2,3,4))).abs().max()):.2e} var-min {float(yr.var(dim=(0,2,3,4), unbiased=False).min()):.2e}"
        if nm.endswith(".c"):
            b = nm.rsplit(".branch2.c", 1)[0]
            line += f"  block-out rel {rel(z, blk_out[b]):.3e}"
        print(line)
    _Unit.trace = None
    for p_, (a, b) in enumerate(zip(fo, fr)):
        print(f"final pathway{p_}: {rel(a, b):.3e}")

run("i3d", "tiny", 8, 8, 2, 32)
run("slowfast", 50, 64, 32, 2, 64)
run("slowfast", 50, 64, 32, 2, 96)
