"""Optimizer-state and checkpoint compatibility with the reference's trainer, without a GPU.

The reference saves `torch.optim.Adam(mdl.parameters()).state_dict()` (`main_dist.py:50`,
`utils/trn_utils.py:699-716`): indices run over EVERY parameter of the reference model in registration
order -- including frozen ones (`embed_tokens.weight`, `fseq_txdec.py`) and the upstream
`sf_mdl.head.projection.{weight,bias}` that `VideoTrunk` never builds -- and only parameters that received a
gradient own a state entry.  For every selector row: a state produced by stock Adam over that parameter list
loads into `ArenaAdam`, comes back out identical, and loads into stock Adam again."""
import pytest
import torch

from vidsitu_amd import checkpoint, synth_data
from vidsitu_amd.extended_config import get_cfg
from vidsitu_amd.mdl_selector import get_mdl_loss_eval
from vidsitu_amd.optim import ArenaAdam, ParamArena, reference_param_order

ROWS = [
    ("vb", "sf_base", {"mdl.sf_mdl_name": "i3d_tiny"}),
    ("vb", "sf_base", {"mdl.sf_mdl_name": "slow_fast_mini"}),
    ("vb", "sf_base_txenc", {"mdl.sf_mdl_name": "slow_fast_mini", "tx_dec.encoder_layers": 2}),
    ("vb_arg", "tx_only", {}),
    ("vb_arg", "new_gpt2_only", {}),
    ("vb_arg", "sfpret_txed_vbarg", {}),
    ("vb_arg", "sfpret_txe_txd_vbarg", {"mdl.tx_enc_type": "new"}),
    ("vb_arg", "sfpret_txe_txd_vbarg", {"mdl.tx_enc_type": "old"}),
    ("vb_arg", "sfpret_txe_txd_vbarg", {"mdl.tx_enc_type": "new_conc"}),
]
SMALL = {"synth.num_verbs": 23, "synth.gpt2_vocab": 211, "mdl.tx_dec_type": "txdec"}


def _model(task, name, extra):
    kw = dict(SMALL)
    kw.update({"task_type": task, "mdl.mdl_name": name})
    kw.update(extra)
    if name == "new_gpt2_only":
        kw["mdl.tx_dec_type"] = "gpt2"
    cfg = get_cfg(kw)
    comm = synth_data.make_comm(cfg)
    torch.manual_seed(0)
    return get_mdl_loss_eval(cfg)["mdl"](cfg=cfg, comm=comm)


def _reference_param_list(mdl):
    """Stand-ins, in the reference's order, for the parameters a reference model would hand to Adam."""
    index = reference_param_order(mdl)
    shapes = {}
    for mname, m in mdl.named_modules():
        if callable(getattr(m, "reference_only_params", None)):
            for suffix, shape in m.reference_only_params():
                shapes[(mname + "." if mname else "") + suffix] = shape
    plist = []
    for n, p in index:
        if p is None:
            plist.append(torch.nn.Parameter(torch.zeros(shapes[n])))  # never receives a gradient
        else:
            q = torch.nn.Parameter(p.detach().clone().contiguous(), requires_grad=p.requires_grad)
            plist.append(q)
    return index, plist


@pytest.mark.parametrize("task,name,extra", ROWS, ids=[f"{r[1]}-{i}" for i, r in enumerate(ROWS)])
def test_adam_state_round_trips_with_stock_adam(task, name, extra):
    mdl = _model(task, name, extra)
    index, plist = _reference_param_list(mdl)
    n_phantom = sum(p is None for _, p in index)
    n_frozen = sum(p is not None and not p.requires_grad for _, p in index)
    if "sf_mdl" in dict(mdl.named_children()):
        assert n_phantom == 2
    ref_opt = torch.optim.Adam(plist, lr=1e-4, betas=(0.9, 0.99))
    g = torch.Generator().manual_seed(1)
    for (_, p), q in zip(index, plist):
        if p is not None and p.requires_grad:
            q.grad = torch.randn(q.shape, generator=g)
    ref_opt.step()
    ref_opt.step()
    sd = ref_opt.state_dict()
    assert len(sd["param_groups"][0]["params"]) == len(index)
    assert len(sd["state"]) == len(index) - n_phantom - n_frozen

    arena = ParamArena(mdl, adopt_conv=False)
    opt = ArenaAdam(arena, lr=3e-3)
    opt.load_state_dict(sd)
    assert int(opt.t) == 2 and opt.lr == 1e-4 and tuple(opt.betas) == (0.9, 0.99)
    back = opt.state_dict()
    assert back["param_groups"][0]["params"] == sd["param_groups"][0]["params"]
    assert set(back["state"]) == set(sd["state"])
    for k, st in sd["state"].items():
        assert torch.equal(back["state"][k]["exp_avg"], st["exp_avg"]), index[k][0]
        assert torch.equal(back["state"][k]["exp_avg_sq"], st["exp_avg_sq"]), index[k][0]
        assert float(back["state"][k]["step"]) == float(st["step"])
    # ... and the reference's optimizer accepts what we wrote
    ref2 = torch.optim.Adam(plist, lr=1.0)
    ref2.load_state_dict(back)
    assert ref2.param_groups[0]["lr"] == 1e-4
    # a state indexed over only the parameters this build constructs is accepted as well
    if n_phantom:
        real = [i for i, (_, p) in enumerate(index) if p is not None]
        remap = {old: new for new, old in enumerate(real)}
        own = {"state": {remap[k]: v for k, v in sd["state"].items()},
               "param_groups": [dict(sd["param_groups"][0], params=list(range(len(real))))]}
        opt2 = ArenaAdam(arena)
        opt2.load_state_dict(own)
        assert torch.equal(opt2.m, opt.m) and torch.equal(opt2.v, opt.v)
    bad = {"state": {}, "param_groups": [dict(sd["param_groups"][0], params=list(range(len(index) + 3)))]}
    with pytest.raises(ValueError):
        opt.load_state_dict(bad)


def test_reference_shaped_checkpoint_loads_strictly_and_saves_the_upstream_keys(tmp_path):
    """A reference SFBase checkpoint carries `sf_mdl.head.projection.*` and a
    `module.` prefix when written under DDP: strict load succeeds, and a file written here contains those keys
    so the reference's strict load of it does too."""
    from oracle.slowfast_ref import SFBaseRef, randomize_bn

    cfg = get_cfg({"mdl.sf_mdl_name": "slow_fast_mini", "synth.num_verbs": 23})
    comm = synth_data.make_comm(cfg)
    mdl = get_mdl_loss_eval(cfg)["mdl"](cfg=cfg, comm=comm)
    ref = SFBaseRef(cfg.sf_mdl, 23)
    randomize_bn(ref, 2)
    sd = {k: v.clone() for k, v in ref.state_dict().items()}
    width = sum(mdl.sf_mdl.dim_out)
    sd["sf_mdl.head.projection.weight"] = torch.randn(400, width)  # the upstream key set
    sd["sf_mdl.head.projection.bias"] = torch.randn(400)
    path = tmp_path / "ref.pth"
    torch.save({"model_state_dict": {"module." + k: v for k, v in sd.items()}, "num_it": 7, "num_epoch": 1,
                "best_met": 0.25, "cfgtxt": "{}"}, path)
    got = checkpoint.load_model_dict(str(path), mdl, strict=True)
    assert got == {"num_it": 7, "num_epoch": 1, "best_met": 0.25}
    for k, v in ref.state_dict().items():
        assert torch.equal(mdl.state_dict()[k], v), k
    out = tmp_path / "models" / "ours.pth"
    checkpoint.save_model_dict(str(out), mdl, None, num_it=8)
    saved = torch.load(out, weights_only=True)["model_state_dict"]
    assert set(saved) == set(sd)
    assert tuple(saved["sf_mdl.head.projection.weight"].shape) == (400, width)
    # wrong upstream shape is an error, not a silent drop
    sd_bad = dict(sd)
    sd_bad["sf_mdl.head.projection.weight"] = torch.zeros(10, 3)
    torch.save({"model_state_dict": sd_bad}, path)
    with pytest.raises(ValueError):
        checkpoint.load_model_dict(str(path), mdl, strict=True)
