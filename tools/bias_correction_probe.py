"""CPU experiment (fp32 torch oracle): how much of the bf16 weight-rounding error of the eval path is a per-channel
CONSTANT that the folded BatchNorm shift can absorb ("bias correction" of post-training quantisation)?
  y = conv(x, W) ;  conv(x, bf16(W)) = y + conv(x, dW),  dW = bf16(W) - W.
The trimmed head averages the feature map over positions, so what reaches the logits is the position MEAN of every
layer's error: E[conv(x, dW)][co] ~= sum_{taps, ci} dW[co, ci, tap] * mu[ci],  mu = per-channel mean of the layer's input
(measured on CALIBRATION clips, not on the clip that is evaluated).  Correcting  beta' = beta - scale * that  costs nothing
at run time.  Prints the logits error of: bf16 weights; bf16 weights + correction from n calibration clips."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oracle.slowfast_ref import SFBaseRef, randomize_bn
from vidsitu_amd import synth_data
from vidsitu_amd.extended_config import get_cfg

torch.set_num_threads(os.cpu_count())
rb = lambda t: t.to(torch.bfloat16).float()
cfg = get_cfg({"mdl.sf_mdl_name": "slow_fast_nl_r50_8x8", "synth.num_verbs": 1564})
comm = synth_data.make_comm(cfg)
torch.manual_seed(0)
ref = SFBaseRef(cfg.sf_mdl, 1564)
randomize_bn(ref, 1)
with torch.no_grad():
    for lin in (ref.proj_head[0], ref.proj_head[2]):
        lin.weight.normal_(0, 0.05)
ref.eval()


def clip(seed, n_ev=1):
    b = synth_data.synth_batch(cfg, comm, bs=1, n_ev=n_ev, seed=seed)
    return [b["frms_ev_slow_tensor"].flatten(0, 1), b["frms_ev_fast_tensor"].flatten(0, 1)]


test = clip(1234)
ncal = int(next((a.split("=")[1] for a in sys.argv if a.startswith("--cal=")), 2))
cal = clip(999, n_ev=ncal)
convs = [(n, m) for n, m in ref.named_modules() if isinstance(m, torch.nn.Conv3d)]
# the BatchNorm behind each convolution: same parent, name conv -> bn / a -> a_bn / branch1 -> branch1_bn ...
mods = dict(ref.named_modules())


def bn_of(name):
    for cand in (name + "_bn", name.rsplit(".", 1)[0] + ".bn"):
        if cand in mods and isinstance(mods[cand], torch.nn.BatchNorm3d):
            return mods[cand]
    raise KeyError(name)


with torch.no_grad():
    lr = ref(test)
    scale = float(lr.abs().max())
    rel = lambda a: float((a - lr).abs().max()) / scale
    # input-channel means on the calibration clips (fp32 oracle; the HIP path would measure its own activations)
    mu = {}
    hooks = [m.register_forward_hook(lambda mod, i, o, n=n: mu.__setitem__(n, i[0].mean(dim=(0, 2, 3, 4)))) for n, m in convs]
    ref(cal)
    for h in hooks:
        h.remove()
    saved = {n: m.weight.data.clone() for n, m in convs}
    bsaved = {n: bn_of(n).bias.data.clone() for n, _ in convs}
    for n, m in convs:
        m.weight.data = rb(saved[n])
    print(f"bf16 weights, no correction: {rel(ref(test)):.3e}")
    for n, m in convs:
        bn = bn_of(n)
        dW = (m.weight.data - saved[n]).sum(dim=(2, 3, 4))  # [co, ci] (borders ignored: every tap sees the mean)
        bias_err = dW @ mu[n]                                # mean of conv(x, dW) per output channel
        sc = bn.weight.data / torch.sqrt(bn.running_var + bn.eps)
        bn.bias.data = bsaved[n] - sc * bias_err
    print(f"bf16 weights + shift correction from {ncal} calibration clip(s): {rel(ref(test)):.3e}")
    e_same = rel(ref(cal[0][:1].new_tensor(0)) if False else ref(test))
    # the same correction evaluated on another unseen clip
    t2 = clip(4321)
    for n, m in convs:
        m.weight.data = saved[n]
        bn_of(n).bias.data = bsaved[n]
    l2 = ref(t2)
    for n, m in convs:
        m.weight.data = rb(saved[n])
    e0 = float((ref(t2) - l2).abs().max()) / float(l2.abs().max())
    for n, m in convs:
        bn = bn_of(n)
        dW = (m.weight.data - saved[n]).sum(dim=(2, 3, 4))
        sc = bn.weight.data / torch.sqrt(bn.running_var + bn.eps)
        bn.bias.data = bsaved[n] - sc * (dW @ mu[n])
    e1 = float((ref(t2) - l2).abs().max()) / float(l2.abs().max())
    print(f"second unseen clip (seed 4321): no correction {e0:.3e}, corrected {e1:.3e}")
