"""CPU experiment (fp32 torch oracle): where does the RESIDUAL of the calibrated-shift eval path come from on video-like
frames (profiles/parity_eval.json "robustness": median clip 8e-4, worst 1.2e-3)?  bf16 weight rounding + the shift
correction (2 video-like calibration clips) applied to ONE group of layers at a time; the logits error of the worst
evaluation clips per group.  usage: python tools/calib_residual_probe.py [n_eval_clips]"""
import os, sys
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


def vclips(seed, n):
    u8 = synth_data.synth_video_u8_batch(cfg, comm, bs=n, n_ev=1, seed=seed)
    b = synth_data.reference_tensors(u8, cfg, comm)
    return [b["frms_ev_slow_tensor"].flatten(0, 1), b["frms_ev_fast_tensor"].flatten(0, 1)]


n_eval = int(sys.argv[1]) if len(sys.argv) > 1 else 3
evals = [vclips(5000 + 17 * i, 1) for i in range(n_eval)]
cal = vclips(999, 2)
convs = [(n, m) for n, m in ref.named_modules() if isinstance(m, torch.nn.Conv3d)]
mods = dict(ref.named_modules())


def bn_of(name):
    for cand in (name + "_bn", name.rsplit(".", 1)[0] + ".bn"):
        if cand in mods and isinstance(mods[cand], torch.nn.BatchNorm3d):
            return mods[cand]
    raise KeyError(name)


groups = {"stems": lambda n: ".s1." in n and "stem" in n, "fuse": lambda n: "_fuse" in n, "s2": lambda n: ".s2." in n,
          "s3": lambda n: ".s3." in n, "s4": lambda n: ".s4." in n, "s5": lambda n: ".s5." in n, "all": lambda n: True}
with torch.no_grad():
    lrs = [ref(e) for e in evals]
    mu = {}
    hooks = [m.register_forward_hook(lambda mod, i, o, n=n: mu.__setitem__(n, i[0].mean(dim=(0, 2, 3, 4)))) for n, m in convs]
    ref(cal)
    for h in hooks:
        h.remove()
    saved = {n: m.weight.data.clone() for n, m in convs}
    bsaved = {n: bn_of(n).bias.data.clone() for n, _ in convs}
    print("layers:", len(convs), "; first names:", [n for n, _ in convs[:3]])
    for gname, sel in groups.items():
        errs_plain, errs_corr = [], []
        for corrected in (False, True):
            for n, m in convs:
                m.weight.data = saved[n]
                bn_of(n).bias.data = bsaved[n]
            k = 0
            for n, m in convs:
                if not sel("." + n):
                    continue
                k += 1
                m.weight.data = rb(saved[n])
                if corrected:
                    bn = bn_of(n)
                    dW = (m.weight.data - saved[n]).sum(dim=(2, 3, 4))
                    sc = bn.weight.data / torch.sqrt(bn.running_var + bn.eps)
                    bn.bias.data = bsaved[n] - sc * (dW @ mu[n])
            for e, lr in zip(evals, lrs):
                err = float((ref(e) - lr).abs().max()) / float(lr.abs().max())
                (errs_corr if corrected else errs_plain).append(err)
        print(f"{gname:6s} ({k:3d} convs): rounded only {max(errs_plain):.2e} (per clip {', '.join(f'{x:.1e}' for x in errs_plain)}) | "
              f"+ calibrated shifts {max(errs_corr):.2e} (per clip {', '.join(f'{x:.1e}' for x in errs_corr)})", flush=True)
