"""Bitwise reproducibility of the training step's gradients under GPU contention: N copies of this process share
cuda:0 (no process group), each repeats fwd + bwd of the mini SlowFast + TxEncoder model from the same state and
compares every gradient with its first pass.  usage (launcher): python tools/probes/repro_stress.py launch COPIES ITERS"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if sys.argv[1] == "launch":
    import io
    from vidsitu_amd import dist_launch
    copies, iters = int(sys.argv[2]), int(sys.argv[3])
    out, err = io.StringIO(), io.StringIO()
    os.environ["VS_STRESS_ITERS"] = str(iters)
    rc = dist_launch.launch_ranks(copies, [sys.executable, os.path.abspath(__file__), "child"], out=out, err=err,
                                  check_devices=False)
    print(out.getvalue().strip())
    e = [ln for ln in err.getvalue().splitlines() if "STRESS" in ln or "Error" in ln]
    e = [ln[:700] for ln in e]
    print("\n".join(e[:24]))
    sys.exit(rc)

import torch
from vidsitu_amd import synth_data
from vidsitu_amd.extended_config import get_cfg
from vidsitu_amd.mdl_selector import get_mdl_loss_eval
from vidsitu_amd.optim import ArenaAdam, ParamArena
from vidsitu_amd.train_step import TrainStep

rank = int(os.environ.get("RANK", "0"))
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
cfg = get_cfg({"mdl.mdl_name": "sf_base_txenc", "mdl.sf_mdl_name": os.environ.get("VS_STRESS_MODEL", "slow_fast_mini"),
               "synth.num_verbs": 31, "tx_dec.encoder_layers": 2, "tx_dec.dropout": 0.0})
comm = synth_data.make_comm(cfg)
torch.manual_seed(0)
sel = get_mdl_loss_eval(cfg)
mdl = sel["mdl"](cfg=cfg, comm=comm).to(dev).train()
loss_fn = sel["loss"](cfg, comm)
arena = ParamArena(mdl)
opt = ArenaAdam(arena, lr=1e-3)
batch = synth_data.synth_batch(cfg, comm, bs=int(os.environ.get("VS_STRESS_BS", "2")), n_ev=int(os.environ.get("VS_STRESS_NEV", "2")), crop=64, seed=1234, device=dev, dtype=torch.bfloat16)
bufs = {k: v.clone() for k, v in mdl.named_buffers()}
names = {id(q): n for n, q in mdl.named_parameters()}
ts = TrainStep(mdl, loss_fn, arena, opt, batch, world=1, use_dist=False)
first, bad = None, {}
for it in range(int(os.environ.get("VS_STRESS_ITERS", "30"))):
    for k, v in mdl.named_buffers():
        v.copy_(bufs[k])
    ts.fwd_bwd()
    torch.cuda.synchronize()
    g = arena.grad.clone()
    if first is None:
        first = g
        continue
    if not torch.equal(g, first):
        for q, off in zip(arena.params, arena.offsets):
            d = float((g[off:off + q.numel()] - first[off:off + q.numel()]).abs().max())
            if d > 0:
                bad.setdefault(names[id(q)], []).append(d)
                if sum(len(v) for v in bad.values()) <= 6:
                    a, b = g[off:off + q.numel()].view(q.shape[0], -1), first[off:off + q.numel()].view(q.shape[0], -1)
                    ne = (a != b)
                    rows_ = ne.any(1).nonzero().flatten()
                    cols_ = ne.any(0).nonzero().flatten()
                    sys.stderr.write(f"STRESSD rank {rank} it {it} {names[id(q)]} {tuple(a.shape)}: {int(ne.sum())} elements differ; "
                                     f"rows {rows_[:6].tolist()}..{rows_[-3:].tolist()} ({rows_.numel()}), cols "
                                     f"{cols_[:6].tolist()}..{cols_[-3:].tolist()} ({cols_.numel()}); got {a[ne][:4].tolist()} want {b[ne][:4].tolist()}\n")
sys.stderr.write(f"STRESS rank {rank}: {sum(len(v) for v in bad.values())} tensor mismatches; " +
                 ", ".join(f"{k} x{len(v)} max {max(v):.2e}" for k, v in sorted(bad.items())[:8]) + "\n")
if rank == 0:
    print("STRESS done")
