"""The N > 1 path on CPU: world_size-2 `gloo` runs of the flat-arena data-parallel step
(vidsitu_amd/optim.py) -- the same code the GPU ranks run over RCCL, minus the HIP kernels.

Checked: parameters/gradients really are views of one buffer (channels-last conv weights
included), rank 0's parameters reach every rank, one all-reduce sums every gradient, and two
ranks on half batches walk exactly the trajectory of one process on the full batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn

from vidsitu_amd.optim import ParamArena
from vidsitu_amd.trunk import Conv3dP


class _Net(nn.Module):
    """A conv with the trunk's channels-last parameter layout + a linear head (no BN, so the
    full-batch and sharded runs are mathematically identical)."""

    def __init__(self):
        super().__init__()
        self.conv = Conv3dP(8, 8, (1, 3, 3), (1, 1, 1), (0, 1, 1))
        self.fc = nn.Linear(8, 5)

    def forward(self, x):
        y = torch.nn.functional.conv3d(x, self.conv.weight, padding=(0, 1, 1)).relu()
        return self.fc(y.mean(dim=(2, 3, 4)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _data():
    g = torch.Generator().manual_seed(0)
    return torch.randn(8, 8, 2, 6, 6, generator=g), torch.randint(0, 5, (8,), generator=g)


def _train(model, arena, x, y, steps, world, bf16=False):
    opt = torch.optim.Adam(arena.params, lr=1e-2, betas=(0.9, 0.99))
    for _ in range(steps):
        arena.zero_grad()
        loss = nn.functional.cross_entropy(model(x), y)
        loss.backward()
        if bf16:
            # bf16 bucket payload (train_step.TrainStep(grad_bf16=True)): pack -> all-reduce the bf16
            # image bucket by bucket -> the optimizer reads the summed image against its fp32 master
            works = []
            for lo, hi in arena.bucket_ranges([[model.fc]]):
                arena.pack_grad_bf16(lo, hi)
                works.append(arena.all_reduce_range(lo, hi, async_op=True, bf16=True, packed=True))
            for wk in works:
                if wk is not None:
                    wk.wait()
            if world == 1:  # single process: no process group, the image is just the rounded gradient
                assert all(wk is None for wk in works)
            arena.unpack_grad_bf16(0, arena.numel)
            w = world
        else:
            w = arena.all_reduce()
            assert w == world
        arena.grad.div_(w)  # the GPU path folds this into vs_adam_step(grad_scale)
        opt.step()
    return arena.data.clone()


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)  # deliberately different initial weights per rank
    model = _Net()
    arena = ParamArena(model, adopt_conv=False)
    arena.broadcast_params(0)
    x, y = _data()
    shard = slice(rank * 4, rank * 4 + 4)
    out = _train(model, arena, x[shard], y[shard], 3, world)
    if rank == 0:
        ret["sharded"] = out
    # the same trajectory with the bf16 gradient transport
    torch.manual_seed(100 + rank)
    model16 = _Net()
    arena16 = ParamArena(model16, adopt_conv=False)
    arena16.broadcast_params(0)
    out16 = _train(model16, arena16, x[shard], y[shard], 3, world, bf16=True)
    assert arena16.grad16 is not None and arena16.grad16.dtype == torch.bfloat16
    g16 = [torch.empty_like(out16) for _ in range(world)]
    dist.all_gather(g16, out16)
    assert all(torch.equal(g16[0], t) for t in g16), "ranks diverged under the bf16 transport"
    if rank == 0:
        ret["sharded_bf16"] = out16
    gathered = [torch.empty_like(out) for _ in range(world)]
    dist.all_gather(gathered, out)
    assert all(torch.equal(gathered[0], g) for g in gathered), "ranks diverged"
    # bucketed, asynchronous all-reduce (bench.py's overlapped step) == the single all-reduce
    ranges = arena.bucket_ranges([[model.fc]])  # -> [fc bucket, everything else]
    assert sorted(ranges) == [(0, arena.offsets[1]), (arena.offsets[1], arena.numel)]
    g = torch.Generator().manual_seed(7 + rank)
    arena.grad.copy_(torch.randn(arena.numel, generator=g))
    want = arena.grad.clone()
    dist.all_reduce(want)
    works = [arena.all_reduce_range(lo, hi, async_op=True) for lo, hi in ranges]
    for wk in works:
        wk.wait()
    assert torch.equal(arena.grad, want)
    dist.destroy_process_group()


def test_arena_views_and_layout():
    model = _Net()
    arena = ParamArena(model, adopt_conv=False)
    w = model.conv.weight
    assert w.permute(0, 2, 3, 4, 1).is_contiguous() and w.grad.permute(0, 2, 3, 4, 1).is_contiguous()
    lo, hi = arena.data.data_ptr(), arena.data.data_ptr() + 4 * arena.numel
    for p in arena.params:
        assert lo <= p.data_ptr() < hi and p.grad is not None
    arena.data.fill_(2.0)
    assert float(model.fc.bias.sum()) == 10.0
    arena.grad.fill_(1.0)
    model.fc.weight.grad = None  # something replaced a gradient: zero_grad re-attaches the view
    arena.zero_grad()
    assert model.fc.weight.grad is not None and float(arena.grad.abs().sum()) == 0.0
    model.fc.weight.grad.add_(1.0)
    assert float(arena.grad.sum()) == model.fc.weight.numel()


@pytest.mark.timeout(300)
def test_two_rank_gloo_matches_single_process():
    torch.manual_seed(100)
    ref_model = _Net()
    ref_arena = ParamArena(ref_model, adopt_conv=False)
    x, y = _data()
    ref_init = ref_arena.data.clone()
    want = _train(ref_model, ref_arena, x, y, 3, 1)
    mgr = mp.Manager()
    ret = mgr.dict()
    for attempt in range(2):  # a rendezvous on a just-freed port can lose a race on a loaded host: one retry
        try:
            mp.spawn(_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
            break
        except Exception as e:  # noqa: BLE001
            rendezvous = any(s in str(e) for s in ("connect", "Connection", "timed out", "Address already in use",
                                                   "store", "Socket"))
            if attempt == 1 or not rendezvous:
                raise
    got = ret["sharded"]
    # mean over 8 = mean of the two 4-sample means; only fp32 summation order differs
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-6), float((got - want).abs().max())
    # bf16 gradient payload: every element of the summed gradient carries <= 2^-8 relative rounding (each
    # rank's image, then the bf16 sum), so after 3 Adam steps at lr 1e-2 the parameters sit within a few
    # lr-sized steps of the fp32 trajectory -- and far closer than the distance travelled
    got16 = ret["sharded_bf16"]
    travelled = float((want - ref_init).abs().max())
    err16 = float((got16 - want).abs().max())
    assert err16 < 0.15 * travelled and err16 < 3e-3, (err16, travelled)
    assert not torch.equal(got16, got)  # the transport really was different
