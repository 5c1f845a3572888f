"""Checkpoint round trip in the reference trainer's file format (SURVEY.md 8f row f4;
`utils/trn_utils.py:631-716`): resume == uninterrupted training bit for bit, the optimizer state loads
into a stock `torch.optim.Adam` and steps to the same parameters, `module.`-prefixed files load."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(dev):
    from vidsitu_amd import synth_data
    from vidsitu_amd.extended_config import get_cfg
    from vidsitu_amd.mdl_selector import get_mdl_loss_eval
    from vidsitu_amd.optim import ArenaAdam, ParamArena

    cfg = get_cfg({"mdl.mdl_name": "sf_base_txenc", "mdl.sf_mdl_name": "slow_fast_mini", "synth.num_verbs": 31,
                   "tx_dec.encoder_layers": 2, "tx_dec.dropout": 0.0, "tx_dec.attention_dropout": 0.0})
    comm = synth_data.make_comm(cfg)
    sel = get_mdl_loss_eval(cfg)
    torch.manual_seed(0)
    mdl = sel["mdl"](cfg=cfg, comm=comm).to(dev).train()
    arena = ParamArena(mdl)
    opt = ArenaAdam(arena, lr=3e-4, betas=(0.9, 0.99))
    batch = synth_data.synth_batch(cfg, comm, bs=2, n_ev=2, crop=32, device=dev, dtype=torch.bfloat16)
    loss_fn = sel["loss"](cfg, comm)

    def step():
        opt.zero_grad()
        loss = loss_fn(mdl(batch), batch)["loss"]
        loss.backward()
        opt.step()
        return float(loss.detach())

    return mdl, arena, opt, step


def test_resume_is_bitwise_uninterrupted_training(dev, tmp_path):
    from vidsitu_amd import checkpoint

    mdl_a, arena_a, _, step_a = _setup(dev)
    losses_a = [step_a() for _ in range(3)]
    mdl_b, _, opt_b, step_b = _setup(dev)
    losses_b = [step_b() for _ in range(2)]
    path = str(tmp_path / "models" / "run.pth")
    checkpoint.save_model_dict(path, mdl_b, opt_b, num_it=2, num_epoch=0, best_met=0.5)
    mdl_c, arena_c, opt_c, step_c = _setup(dev)
    with torch.no_grad():  # a different starting point: the load must overwrite everything
        arena_c.data.mul_(0.5)
    got = checkpoint.load_model_dict(path, mdl_c, opt_c, load_opt=True)
    assert got == {"num_it": 2, "num_epoch": 0, "best_met": 0.5}
    losses_c = [step_c()]
    assert losses_b == losses_a[:2] and losses_c[0] == losses_a[2]
    assert torch.equal(arena_c.data, arena_a.data)
    for (ka, va), (kc, vc) in zip(mdl_a.state_dict().items(), mdl_c.state_dict().items()):
        assert ka == kc and torch.equal(va, vc), ka
    assert checkpoint.load_model_dict(str(tmp_path / "missing.pth"), mdl_c) is None


def test_optimizer_state_is_torch_adam_format_and_module_prefix_loads(dev, tmp_path):
    from vidsitu_amd import checkpoint

    mdl, arena, opt, step = _setup(dev)
    step(), step()
    path = str(tmp_path / "run.pth")
    ckpt = checkpoint.save_model_dict(path, mdl, opt, num_it=2)
    # (1) a stock Adam over the reference's parameter list -- this model's parameters in `mdl.parameters()`
    # order plus the upstream `sf_mdl.head.projection.{weight,bias}` the reference model owns and never trains
    # (optim.reference_param_order) -- accepts the state and makes the same third step
    from vidsitu_amd.checkpoint import reference_only_keys
    from vidsitu_amd.optim import reference_param_order

    index = reference_param_order(mdl)
    ph_shapes = reference_only_keys(mdl)
    assert [n for n, p in index if p is None] == ["sf_mdl.head.projection.weight", "sf_mdl.head.projection.bias"]
    all_params = [torch.nn.Parameter(torch.zeros(ph_shapes[n], dtype=torch.float64)) if p is None else
                  torch.nn.Parameter(p.detach().clone().contiguous().cpu().double()) for n, p in index]
    params = [q for q, (_, p) in zip(all_params, index) if p is not None]
    assert len(params) == len(arena.params)
    adam = torch.optim.Adam(all_params, lr=3e-4, betas=(0.9, 0.99))
    sd = ckpt["optimizer_state_dict"]
    sd64 = {"state": {k: {n: (t.double() if torch.is_tensor(t) and t.dim() > 0 else t) for n, t in st.items()}
                      for k, st in sd["state"].items()}, "param_groups": sd["param_groups"]}
    adam.load_state_dict(sd64)
    opt.zero_grad()
    step()  # the HIP path's third step (its gradients stay in the arena)
    for p, q in zip(params, arena.params):
        p.grad = q.grad.detach().clone().contiguous().cpu().double()
    adam.step()
    worst = max(float((p.detach() - q.detach().cpu().double()).abs().max()) for p, q in zip(params, arena.params))
    assert worst < 1e-6, worst
    # (2) a file written by a DistributedDataParallel-wrapped model of the reference
    ckpt2 = torch.load(path, weights_only=True)
    ckpt2["model_state_dict"] = {"module." + k: v for k, v in ckpt2["model_state_dict"].items()}
    path2 = str(tmp_path / "ddp.pth")
    torch.save(ckpt2, path2)
    mdl2, arena2, opt2, _ = _setup(dev)
    checkpoint.load_model_dict(path2, mdl2, opt2, load_opt=True)
    sd2 = mdl2.state_dict()
    for ka, va in ckpt["model_state_dict"].items():
        if ka in ph_shapes:  # placeholders of the upstream head (never built here)
            assert tuple(va.shape) == ph_shapes[ka]
            continue
        assert torch.equal(va, sd2[ka].cpu()), ka
    assert set(sd2) == set(ckpt["model_state_dict"]) - set(ph_shapes)
