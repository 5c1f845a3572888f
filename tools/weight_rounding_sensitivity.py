"""Which convolutions' bf16 weight rounding separates the eval path from north_star's "logits within 1e-3"?
CPU only (the fp32 torch oracle, oracle/slowfast_ref.py): one 224^2 SlowFast-R50 clip, the 1564-verb head, the clip and
weights of tests/test_gpu_parity_full.py::test_slowfast_r50_one_clip_224_eval_logits.  For every Conv3d l: logits with
ONLY l's weights rounded to bf16 against the fp32 logits (max |diff| / max |logit|), next to the layer's share of the
forward MACs (what splitting it, W = W_hi + W_lo, costs).  Writes tests/golden/weight_rounding_sensitivity.json.
usage: python tools/weight_rounding_sensitivity.py [--check name1,name2,...]   (--check: everything rounded EXCEPT the
named layers, which keep W_hi + W_lo: the error a selective split leaves)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oracle.slowfast_ref import SFBaseRef, randomize_bn, count_conv_macs_params
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
batch = synth_data.synth_batch(cfg, comm, bs=1, n_ev=1, seed=1234)
inp = [batch["frms_ev_slow_tensor"].flatten(0, 1), batch["frms_ev_fast_tensor"].flatten(0, 1)]
convs = [(n, m) for n, m in ref.named_modules() if isinstance(m, torch.nn.Conv3d)]
macs = {}
hooks = []
for n, m in convs:
    def hook(mod, i, o, n=n):
        macs[n] = o.numel() * mod.weight[0].numel()
    hooks.append(m.register_forward_hook(hook))
with torch.no_grad():
    t0 = time.time(); lr = ref(inp); t1 = time.time()
for h in hooks:
    h.remove()
scale = float(lr.abs().max())
tot = sum(macs.values())
print(f"{len(convs)} convolutions, {tot / 1e9:.2f} GMAC, fp32 forward {t1 - t0:.1f} s, max |logit| {scale:.3f}", flush=True)
rel = lambda a: float((a - lr).abs().max()) / scale

chk = next((a.split("=")[1] for a in sys.argv if a.startswith("--check=")), None)
if chk is not None:
    keep = set(chk.split(",")) if chk else set()
    saved = {n: m.weight.data.clone() for n, m in convs}
    with torch.no_grad():
        for n, m in convs:
            w = saved[n]
            m.weight.data = rb(w) + rb(w - rb(w)) if n in keep else rb(w)
        e = rel(ref(inp))
    cost = sum(macs[n] for n in keep) / tot
    print(f"split {len(keep)} layers ({100 * cost:.1f} % of the MACs): weight-rounding error left {e:.3e}")
    sys.exit(0)

out = []
with torch.no_grad():
    for i, (n, m) in enumerate(convs):
        w = m.weight.data
        m.weight.data = rb(w)
        e = rel(ref(inp))
        m.weight.data = w
        out.append({"layer": n, "err": e, "mac_share": macs[n] / tot})
        print(f"{i:3d} {n:40s} err {e:.3e}  macs {100 * macs[n] / tot:5.2f} %", flush=True)
    saved = {n: m.weight.data.clone() for n, m in convs}
    for n, m in convs:
        m.weight.data = rb(saved[n])
    e_all = rel(ref(inp))
    for n, m in convs:
        m.weight.data = saved[n]
print(f"all layers rounded: {e_all:.3e}; sum of the single-layer errors {sum(o['err'] for o in out):.3e}")
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "weight_rounding_sensitivity.json")
json.dump({"all_rounded": e_all, "scale": scale, "layers": out,
           "source": "tools/weight_rounding_sensitivity.py (fp32 torch oracle, CPU; clip seed 1234, weights seed 0)"},
          open(dst, "w"), indent=1)
print("wrote", dst)
