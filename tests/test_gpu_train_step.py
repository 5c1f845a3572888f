"""The single-process training step of `vidsitu_amd.train_step.TrainStep` (`utils/trn_utils.py:590-615`:
zero_grad -> forward -> loss -> backward -> optimizer.step): the variant that starts Adam early -- one tick of the
step count, then the update of each gradient range on a side stream as soon as its backward segment is done --
must leave bitwise the gradients, parameters, moments and step count of the plain step (one Adam launch at the
end), eager and replayed from a hipGraph."""
import gc

import pytest
import torch

pytestmark = pytest.mark.gpu


def _poison(arena):
    """NaN in every parameter's gradient; the alignment gaps between parameters stay zero (nothing writes them
    after the arena's construction, and the optimizer runs over the whole arena)."""
    arena.grad.fill_(float("nan"))
    for off, nxt, p in zip(arena.offsets, arena.offsets[1:], arena.params):
        arena.grad[off + p.numel():nxt].zero_()


def test_ranged_adam_beside_the_backward_pass_is_bitwise_the_plain_step(dev):
    from vidsitu_amd import synth_data
    from vidsitu_amd.extended_config import get_cfg
    from vidsitu_amd.mdl_selector import get_mdl_loss_eval
    from vidsitu_amd.optim import ArenaAdam, ParamArena
    from vidsitu_amd.train_step import TrainStep

    cfg = get_cfg({"mdl.mdl_name": "sf_base_txenc", "mdl.sf_mdl_name": "slow_fast_mini", "synth.num_verbs": 31,
                   "tx_dec.encoder_layers": 2, "tx_dec.dropout": 0.0})
    comm = synth_data.make_comm(cfg)
    torch.manual_seed(0)
    sel = get_mdl_loss_eval(cfg)
    mdl = sel["mdl"](cfg=cfg, comm=comm).to(dev).train()
    loss_fn = sel["loss"](cfg, comm)
    arena = ParamArena(mdl)
    opt = ArenaAdam(arena, lr=1e-3)
    batch = synth_data.synth_batch(cfg, comm, bs=2, n_ev=2, crop=64, device=dev, dtype=torch.bfloat16)
    init = arena.data.clone()
    bufs = {k: v.clone() for k, v in mdl.named_buffers()}

    def reset():
        arena.data.copy_(init)
        opt.m.zero_(); opt.v.zero_(); opt.t.zero_()
        for k, v in mdl.named_buffers():
            v.copy_(bufs[k])
        arena.refresh()
        torch.cuda.synchronize()

    def state():
        return [t.clone() for t in (arena.grad, arena.data, opt.m, opt.v, opt.t)]

    def run(ts):
        gc.collect()
        torch.cuda.synchronize()
        reset()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ts.step()
            ts.step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        eager = state()
        ts.capture()
        reset()
        for _ in range(2):
            _poison(arena)
            ts.replay()
        torch.cuda.synchronize()
        replayed = state() + [ts.loss.clone()]
        ts.graphs = None
        return eager, replayed

    plain = TrainStep(mdl, loss_fn, arena, opt, batch, adam_overlap=False)
    assert not plain.adam_overlap and len(plain.segments) == 1
    e0, r0 = run(plain)
    early = TrainStep(mdl, loss_fn, arena, opt, batch, adam_overlap=True)
    assert early.adam_overlap and len(early.segments) == 4
    lo_hi = sorted(r for _, r in early.segments)
    assert lo_hi[0][0] == 0 and lo_hi[-1][1] == arena.numel and all(a[1] == b[0] for a, b in zip(lo_hi, lo_hi[1:]))
    e1, r1 = run(early)
    assert int(r0[4]) == 2 and not torch.equal(r0[1], init) and torch.isfinite(r0[0]).all()
    for name, a, b in zip(("grad", "param", "exp_avg", "exp_avg_sq", "step", "loss"), r0, r1):
        assert torch.equal(a, b), f"replayed {name} differs"
    for name, a, b in zip(("grad", "param", "exp_avg", "exp_avg_sq", "step"), e0, e1):
        assert torch.equal(a, b), f"eager {name} differs"
    for name, a, b in zip(("grad", "param", "exp_avg", "exp_avg_sq", "step"), e0, r0):
        assert torch.equal(a, b), f"eager vs replayed {name} differs"


def test_gradient_fill_is_skipped_only_where_the_backward_overwrites(dev):
    """`TrainStep(grad_fill="learn")` (bench.py; the default is a full fill every step): the first eager step fills the gradient arena and learns which parameters get
    their gradient through autograd's AccumulateGrad; later steps zero only those.  (a) the HIP model: none is
    learned, and steps from NaN-poisoned gradients are bitwise the steps of a TrainStep that fills every time,
    eager and replayed; (b) a model with torch-native parameters: all of them are learned and the trajectory is
    the filled one as well (without the zeroing their gradients would pile up step after step)."""
    from vidsitu_amd import synth_data
    from vidsitu_amd.extended_config import get_cfg
    from vidsitu_amd.mdl_selector import get_mdl_loss_eval
    from vidsitu_amd.optim import ArenaAdam, ParamArena
    from vidsitu_amd.train_step import TrainStep

    def trajectory(mdl, loss_fn, batch, grad_fill, graph):
        arena = ParamArena(mdl)
        opt = ArenaAdam(arena, lr=1e-3)
        ts = TrainStep(mdl, loss_fn, arena, opt, batch, grad_fill=grad_fill, adam_overlap=False)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ts.step()
            if graph:
                ts.capture()
            for _ in range(2):
                _poison(arena)
                ts.run()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        return ts, [arena.grad.clone(), arena.data.clone(), opt.m.clone(), opt.v.clone(), ts.loss.clone()]

    # (a) SlowFast mini + TxEncoder, no dropout (two model instances from one seed: same weights)
    def hip_model():
        cfg = get_cfg({"mdl.mdl_name": "sf_base_txenc", "mdl.sf_mdl_name": "slow_fast_mini", "synth.num_verbs": 31,
                       "tx_dec.encoder_layers": 2, "tx_dec.dropout": 0.0})
        comm = synth_data.make_comm(cfg)
        torch.manual_seed(0)
        sel = get_mdl_loss_eval(cfg)
        mdl = sel["mdl"](cfg=cfg, comm=comm).to(dev).train()
        batch = synth_data.synth_batch(cfg, comm, bs=2, n_ev=2, crop=64, device=dev, dtype=torch.bfloat16)
        return mdl, sel["loss"](cfg, comm), batch

    for graph in (False, True):
        _, ref = trajectory(*hip_model(), True, graph)
        ts, got = trajectory(*hip_model(), "learn", graph)
        assert ts._accumulated == [], [tuple(p.shape) for p in ts._accumulated]
        assert torch.isfinite(got[0]).all()
        for name, a, b in zip(("grad", "param", "exp_avg", "exp_avg_sq", "loss"), ref, got):
            assert torch.equal(a, b), f"graph={graph}: {name} differs without the per-step fill"

    # (b) torch-native parameters
    class Plain(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a, self.b = torch.nn.Linear(16, 32), torch.nn.Linear(32, 4)

        def forward(self, batch):
            return self.b(torch.tanh(self.a(batch["x"])))

    def plain_model():
        torch.manual_seed(1)
        mdl = Plain().to(dev)
        batch = {"x": torch.randn(8, 16, device=dev), "y": torch.randint(0, 4, (8,), device=dev)}
        return mdl, (lambda out, b: {"loss": torch.nn.functional.cross_entropy(out, b["y"])}), batch

    _, ref = trajectory(*plain_model(), True, False)
    ts, got = trajectory(*plain_model(), "learn", False)
    assert len(ts._accumulated) == 4
    for name, a, b in zip(("grad", "param", "exp_avg", "exp_avg_sq", "loss"), ref, got):
        assert torch.equal(a, b), f"torch-native model: {name} differs"


def test_learned_fill_set_raises_when_a_later_step_routes_a_gradient_through_autograd(dev):
    """with grad_fill="learn" the first eager step decides which gradients get zeroed.  A parameter whose
    gradient starts arriving through AccumulateGrad on a LATER step (here: a module switched from a path that writes
    p.grad itself to plain autograd) must not silently accumulate -- the step raises, eagerly and at capture; with the
    default grad_fill=True the same switch is harmless (trajectory == a model that used autograd from the start)."""
    from vidsitu_amd.optim import ArenaAdam, ParamArena
    from vidsitu_amd.train_step import TrainStep

    class WritesOwnGrad(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w):
            ctx.save_for_backward(x, w)
            return x @ w.t()

        @staticmethod
        def backward(ctx, dy):
            x, w = ctx.saved_tensors
            w.grad.copy_(dy.t() @ x)  # overwrite semantics, like the HIP modules
            return dy @ w, None

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = torch.nn.Linear(16, 4, bias=False)
            self.direct = True

        def forward(self, batch):
            if self.direct:
                return WritesOwnGrad.apply(batch["x"], self.lin.weight)
            return self.lin(batch["x"])

    def make(direct):
        torch.manual_seed(3)
        m = M().to(dev)
        m.direct = direct
        batch = {"x": torch.randn(8, 16, device=dev, requires_grad=True), "y": torch.randint(0, 4, (8,), device=dev)}
        loss = (lambda out, b: {"loss": torch.nn.functional.cross_entropy(out, b["y"])})
        arena = ParamArena(m)
        return m, arena, TrainStep(m, loss, arena, ArenaAdam(arena, lr=1e-2), batch,
                                   grad_fill=make.fill, adam_overlap=False)

    make.fill = "learn"
    m, arena, ts = make(True)
    ts.step()
    assert ts._accumulated == []
    ts.step()
    m.direct = False
    with pytest.raises(RuntimeError, match="lin.weight"):
        ts.step()
    ts_learn = ts
    # default: a full fill every step -- switching paths changes nothing
    make.fill = True
    m, arena, ts = make(True)
    ts.step()
    m.direct = False
    ts.step(); ts.step()
    m2, arena2, ts2 = make(False)
    ts2.step(); ts2.step(); ts2.step()
    torch.cuda.synchronize()
    assert torch.equal(arena.data, arena2.data) and torch.equal(arena.grad, arena2.grad)
    # ... and the pass a hipGraph is captured from is watched too (last: an aborted capture leaves torch's CUDA generator
    # in capture mode, nothing random may follow in this process)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with pytest.raises(RuntimeError, match="lin.weight"):
            ts_learn.capture()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert ts_learn.graphs is None
