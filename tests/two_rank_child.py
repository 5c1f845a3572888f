"""Child of tests/test_gpu_dist_two_ranks.py: ONE of TWO ranks, both on cuda:0, process group "gloo" (the test pool has
one GPU per box, and RCCL refuses two ranks on one device; gloo moves CUDA tensors through host memory, which is all
this test needs of the transport).  What runs is bench.py's / main_dist.py's distributed step
(`vidsitu_amd.train_step.TrainStep`, world = 2): rank r trains on its own clips, the segment graphs are replayed with a
bucket all-reduce behind each, Adam averages by 1 / world.  Checked on rank 0 against a single-process computation of
BOTH ranks' gradients from the same initial state:

  * the all-reduced gradient arena == g(rank 0 clips) + g(rank 1 clips), bit for bit (a two-term fp32 sum);
  * the parameters after the step == one Adam step on that sum with grad_scale 1/2, bit for bit;
  * bf16 payload: the summed bf16 image == bf16(g0) + bf16(g1) rounded once, parameters within the rounding;
  * both ranks hold identical parameters afterwards (all_gather of a checksum).

Reference: `main_dist.py:68-79` (DDP wrap), `utils/dat_utils.py:40-43` (per-rank shard)."""
import gc
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert world == 2
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    from vidsitu_amd import synth_data
    from vidsitu_amd.extended_config import get_cfg
    from vidsitu_amd.mdl_selector import get_mdl_loss_eval
    from vidsitu_amd.optim import ArenaAdam, ParamArena
    from vidsitu_amd.train_step import TrainStep

    sf_name, crop = os.environ.get("VS_TWO_RANK_MODEL", "slow_fast_mini:64").split(":")
    cfg = get_cfg({"mdl.mdl_name": "sf_base_txenc", "mdl.sf_mdl_name": sf_name, "synth.num_verbs": 31,
                   "tx_dec.encoder_layers": 2, "tx_dec.dropout": 0.0})
    comm = synth_data.make_comm(cfg)
    torch.manual_seed(0)
    sel = get_mdl_loss_eval(cfg)
    mdl = sel["mdl"](cfg=cfg, comm=comm).to(dev).train()
    loss_fn = sel["loss"](cfg, comm)
    arena = ParamArena(mdl)
    if rank == 1:  # rank 1 starts from different parameters: the broadcast has to repair that
        arena.data.mul_(1.5)
    arena.broadcast_params(0)
    opt = ArenaAdam(arena, lr=1e-3)
    batches = [synth_data.synth_batch(cfg, comm, bs=2, n_ev=2, crop=int(crop), seed=1234 + r, device=dev,
                                      dtype=torch.bfloat16) for r in range(world)]
    init = arena.data.clone()
    bufs = {k: v.clone() for k, v in mdl.named_buffers()}

    def reset():
        arena.data.copy_(init)
        opt.m.zero_(); opt.v.zero_(); opt.t.zero_()
        for k, v in mdl.named_buffers():
            v.copy_(bufs[k])
        arena.refresh()
        torch.cuda.synchronize()

    def run(ts, graph):
        gc.collect()
        torch.cuda.synchronize()
        reset()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ts.step()  # eager (also the capture warm-up)
            if graph:
                ts.capture()
                torch.cuda.synchronize()
                reset()
                arena.grad.fill_(float("nan"))
                for off, nxt, q in zip(arena.offsets, arena.offsets[1:], arena.params):
                    arena.grad[off + q.numel():nxt].zero_()
                ts.replay()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        out = (arena.grad.clone(), arena.data.clone(), None if arena.grad16 is None else arena.grad16.clone())
        ts.graphs = None
        return out

    # single-process gradients of each rank's clips from the same initial state (no collective: every rank computes both)
    g_single = []
    for r in range(world):
        ts = TrainStep(mdl, loss_fn, arena, opt, batches[r], world=1, use_dist=False)
        reset()
        ts.fwd_bwd()
        torch.cuda.synchronize()
        g_single.append(arena.grad.clone())
    assert float((g_single[0] - g_single[1]).abs().max()) > 0, "the two ranks' clips must differ"
    # the same once more (the other rank's kernels share the GPU and shift every timing): bitwise reproducible
    ts = TrainStep(mdl, loss_fn, arena, opt, batches[0], world=1, use_dist=False)
    reset()
    ts.fwd_bwd()
    torch.cuda.synchronize()
    assert torch.equal(arena.grad, g_single[0]), "single-process gradients are not reproducible"

    def where(a, b):
        rows = []
        names = {id(q): n for n, q in mdl.named_parameters()}
        for q, off in zip(arena.params, arena.offsets):
            name = names[id(q)]
            d = float((a[off:off + q.numel()] - b[off:off + q.numel()]).abs().max())
            if d > 0:
                rows.append((d, name, off))
        rows.sort(reverse=True)
        return f"{len(rows)} parameters differ; worst: " + ", ".join(f"{n}@{o} {d:.2e}" for d, n, o in rows[:6])
    g_sum = g_single[0] + g_single[1]
    # ... and one Adam step on the sum with grad_scale = 1 / world
    reset()
    arena.grad.copy_(g_sum)
    opt.step(world=world)
    torch.cuda.synchronize()
    p_want = arena.data.clone()

    # the transport alone: gloo's sum of the two ranks' single-process gradients
    probe = g_single[rank].clone()
    dist.all_reduce(probe)
    torch.cuda.synchronize()
    assert torch.equal(probe, g_sum), f"rank {rank}: gloo all_reduce of a CUDA tensor != g0 + g1: {where(probe, g_sum)}"
    # the distributed step's LOCAL gradients (collectives off): the deferred, segmented backward on this rank's clips
    for graph in (False, True):
        ts = TrainStep(mdl, loss_fn, arena, opt, batches[rank], world=world, overlap=True, use_dist=True)
        ts.collectives = False
        g, _, _ = run(ts, graph)
        assert torch.equal(g, g_single[rank]), f"rank {rank} graph={graph}: local gradients of the segmented step " \
                                               f"differ from the single-process ones: {where(g, g_single[rank])}"
    dist.barrier()
    if os.environ.get("VS_TWO_RANK_ONLY_LOCAL") == "1":  # tools/probes/two_rank_race.sh
        dist.destroy_process_group()
        if rank == 0:
            print("TWO_RANK_CHILD_OK")
        return

    for graph in (False, True):
        for overlap in (True, False):
            ts = TrainStep(mdl, loss_fn, arena, opt, batches[rank], world=world, overlap=overlap, use_dist=True)
            assert len(ts.segments) == (4 if overlap else 1)
            g, p, _ = run(ts, graph)
            tag = f"graph={graph} overlap={overlap}"
            assert torch.equal(g, g_sum), f"{tag} rank {rank}: all-reduced gradients != g0 + g1, max " \
                                          f"{float((g - g_sum).abs().max()):.3e}; segments {[r for _, r in ts.segments]}; " \
                                          f"{where(g, g_sum)}"
            assert torch.equal(p, p_want), f"{tag}: parameters after the step differ"
    # bf16 payload
    ts = TrainStep(mdl, loss_fn, arena, opt, batches[rank], world=world, overlap=True, use_dist=True, grad_bf16=True)
    g, p, g16 = run(ts, True)
    assert torch.equal(g, g_single[rank])  # the fp32 arena keeps the local gradients
    want16 = (g_single[0].to(torch.bfloat16).float() + g_single[1].to(torch.bfloat16).float()).to(torch.bfloat16)
    assert torch.equal(g16, want16), "bf16 payload: the summed image is not bf16(bf16(g0) + bf16(g1))"
    assert float((p - p_want).abs().max()) < 1e-3 and not torch.equal(p, init)
    # both ranks ended with the same parameters
    chk = torch.stack([p.double().sum(), p.double().abs().sum()]).cpu()
    both = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(both, chk)
    assert torch.equal(both[0], both[1]), "ranks diverged"
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("TWO_RANK_CHILD_OK")


if __name__ == "__main__":
    main()
