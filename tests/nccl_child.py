"""Child process of tests/test_gpu_dist_nccl.py: ONE rank, backend "nccl" (= RCCL on ROCm), the
distributed training step of bench.py (`vidsitu_amd.train_step.TrainStep`) on cuda:0.

World size 1 makes every all-reduce an identity, so the distributed step -- trunk backward deferred out
of autograd, one hipGraph per segment, a bucket all-reduce entered on RCCL's stream behind each segment,
Adam in a last graph -- must leave exactly the gradients and parameters of the single-process step
(one hipGraph, no process group).  Mirrors `main_dist.py:68-79` (DDP wrap) of the reference."""
import gc
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    probe = torch.arange(8, device=dev, dtype=torch.float32)
    dist.all_reduce(probe)  # RCCL communicator really comes up
    torch.cuda.synchronize()
    assert probe.tolist() == list(range(8))

    from vidsitu_amd import synth_data
    from vidsitu_amd.extended_config import get_cfg
    from vidsitu_amd.mdl_selector import get_mdl_loss_eval
    from vidsitu_amd.optim import ArenaAdam, ParamArena
    from vidsitu_amd.train_step import TrainStep

    sf_name, crop = os.environ.get("VS_NCCL_TEST_MODEL", "slow_fast_mini:64").split(":")
    cfg = get_cfg({"mdl.mdl_name": "sf_base_txenc", "mdl.sf_mdl_name": sf_name, "synth.num_verbs": 31,
                   "tx_dec.encoder_layers": 2, "tx_dec.dropout": 0.0})
    comm = synth_data.make_comm(cfg)
    torch.manual_seed(0)
    sel = get_mdl_loss_eval(cfg)
    mdl = sel["mdl"](cfg=cfg, comm=comm).to(dev).train()
    loss_fn = sel["loss"](cfg, comm)
    arena = ParamArena(mdl)
    arena.broadcast_params(0)
    opt = ArenaAdam(arena, lr=1e-3)
    batch = synth_data.synth_batch(cfg, comm, bs=2, n_ev=2, crop=int(crop), device=dev, dtype=torch.bfloat16)
    init = arena.data.clone()
    bufs = {k: v.clone() for k, v in mdl.named_buffers()}

    def reset():
        arena.data.copy_(init)
        opt.m.zero_(); opt.v.zero_(); opt.t.zero_()
        for k, v in mdl.named_buffers():
            v.copy_(bufs[k])
        arena.refresh()
        torch.cuda.synchronize()

    def run(ts, replays=2):
        # graphs of the previous variant must be gone before a new capture starts: a CUDAGraph destroyed by the
        # garbage collector in the middle of a capture is an illegal call on the capturing stream
        gc.collect()
        torch.cuda.synchronize()
        reset()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ts.step()  # eager warm-up (allocator, lane streams) -- also exercises the eager path
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        eager = (arena.grad.clone(), arena.data.clone())
        ts.capture()
        reset()
        for _ in range(replays):
            arena.grad.fill_(float("nan"))
            for off, nxt, q in zip(arena.offsets, arena.offsets[1:], arena.params):  # alignment gaps stay zero
                arena.grad[off + q.numel():nxt].zero_()
            ts.replay()
        torch.cuda.synchronize()
        out = (arena.grad.clone(), arena.data.clone(), ts.loss.clone())
        ts.graphs = None
        return eager, out

    # reference: the single-process step (no collective), one hipGraph
    (ge, _), (g_ref, p_ref, l_ref) = run(TrainStep(mdl, loss_fn, arena, opt, batch, world=1, use_dist=False))
    assert torch.isfinite(g_ref).all() and float(g_ref.abs().max()) > 0 and not torch.equal(p_ref, init)

    # bench.py's distributed step: segment graphs + async bucket all-reduces over RCCL
    ts = TrainStep(mdl, loss_fn, arena, opt, batch, world=1, overlap=True, use_dist=True)
    assert len(ts.segments) == 4 and len({r for _, r in ts.segments}) == 4
    n_graphs = []
    _cap = ts.capture
    ts.capture = lambda: (_cap(), n_graphs.append(len(ts.graphs)))
    (ge, pe), (g, p, l) = run(ts)
    assert n_graphs == [5]
    assert torch.equal(g, g_ref), f"overlapped dist step: gradients differ, max {float((g - g_ref).abs().max()):.3e}"
    assert torch.equal(p, p_ref) and torch.equal(l, l_ref)

    # the same without overlap: one segment, one all-reduce, still replayed from graphs
    ts = TrainStep(mdl, loss_fn, arena, opt, batch, world=1, overlap=False, use_dist=True)
    n_graphs = []
    _cap2 = ts.capture
    ts.capture = lambda: (_cap2(), n_graphs.append(len(ts.graphs)))
    (_, _), (g, p, l) = run(ts)
    assert n_graphs == [2] and torch.equal(g, g_ref) and torch.equal(p, p_ref)

    # bf16 bucket payload: RCCL sums the bf16 image, Adam reads it against the fp32 master.  One replay, so that
    # the gradients are those of the unchanged initial parameters.
    (_, _), (g1_ref, p1_ref, _) = run(TrainStep(mdl, loss_fn, arena, opt, batch, world=1, use_dist=False), replays=1)
    ts = TrainStep(mdl, loss_fn, arena, opt, batch, world=1, overlap=True, use_dist=True, grad_bf16=True)
    (_, _), (g, p, l) = run(ts, replays=1)
    assert torch.equal(g, g1_ref)  # the fp32 arena is untouched by the transport
    assert torch.equal(arena.grad16.float(), g1_ref.to(torch.bfloat16).float())
    # one Adam step (lr 1e-3) from gradients rounded to 8 bits: |update| <= lr, the rounding moves it by a fraction
    diff = float((p - p1_ref).abs().max())
    assert diff < 1e-3, diff
    assert not torch.equal(p, init)

    # a capture failure must raise, never fall back to eager
    class Boom(TrainStep):
        def fwd_bwd(self):
            super().fwd_bwd()
            raise ValueError("stand-in for an operation that cannot be captured")

    bad = Boom(mdl, loss_fn, arena, opt, batch, world=1, overlap=False, use_dist=True)
    bad.segments = [(bad.fwd_bwd, bad.segments[0][1])]
    try:
        bad.capture()
    except RuntimeError as e:
        assert "capture" in str(e)
    else:
        raise AssertionError("a failing capture did not raise")
    assert bad.graphs is None
    torch.cuda.synchronize()
    dist.destroy_process_group()
    print("NCCL_CHILD_OK")


if __name__ == "__main__":
    main()
