#!/usr/bin/env python3
"""`python main_dist.py <uid> --dotted.key=value ...` -- the reference's entry point
(`main_dist.py:132-172`: cfg -> `launch_job` -> `main_fn` -> `learner_init` -> `Learner.fit`)
for the hot path, on synthetic batches (the 50 GB frame dataset is out of scope).

Kept: uid + dotted overrides, `num_gpus` from the visible devices, one process per GPU through
`mp.spawn` with a localhost TCP rendezvous and backend `cfg.DIST_BACKEND` ("nccl" == RCCL on
ROCm; `utils/trn_dist_utils.py:5-42`), `get_mdl_loss_eval` plugin lookup, Adam(betas=(0.9, 0.99))
at `train.lr`, per-rank batch = `train.bs // num_gpus` (`utils/dat_utils.py:42-43`).
`train.resume` / `train.resume_path` / `train.load_opt` / `train.strict_load` restore a checkpoint in the
reference's file format before training and `misc.tmp_path/models/<uid>.pth` is written after it
(`vidsitu_amd/checkpoint.py`; `utils/trn_utils.py:631-716`).
Rendezvous port: `MASTER_PORT` when set, else a free port picked by the parent (the reference
hard-codes 9997, `utils/trn_dist_utils.py:30`, so two jobs on one node collide).  `task_type=vb_arg`
rows train on the SRL batch contract (`synth_data.synth_srl_batch`).  An empty `train.resume_path`
means `<tmp_path>/models/<uid>.pth`, as `Learner.load_model_dict` does (`utils/trn_utils.py:643-646`).
Not kept: MLflow / progress bars / per-epoch checkpoint rotation (`utils/trn_utils.py`, out of scope).
The loop never syncs with the host inside a step (the reference does twice: `trn_utils.py:600,610`).
"""
import os
import socket
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vidsitu_amd import checkpoint, synth_data
from vidsitu_amd.extended_config import get_cfg
from vidsitu_amd.mdl_selector import get_mdl_loss_eval
from vidsitu_amd.optim import ArenaAdam, ParamArena


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def main_fn(rank, cfg, steps, port=None):
    world = cfg.num_gpus
    if cfg.do_dist:
        torch.cuda.set_device(rank)
        dist.init_process_group(backend=cfg.DIST_BACKEND, init_method=f"tcp://127.0.0.1:{port}",
                                world_size=world, rank=rank)
    dev = torch.device("cuda", rank)
    comm = synth_data.make_comm(cfg)
    sel = get_mdl_loss_eval(cfg)
    torch.manual_seed(0)
    mdl = sel["mdl"](cfg=cfg, comm=comm).to(dev)
    loss_fn = sel["loss"](cfg, comm)
    eval_fn = sel["evl"](cfg, comm, dev)
    arena = ParamArena(mdl)
    arena.broadcast_params(0)
    opt = ArenaAdam(arena, lr=cfg.train.lr, betas=(0.9, 0.99))
    num_it = 0
    model_file = f"{cfg.misc.tmp_path}/models/{cfg.uid}.pth"
    if cfg.train.resume:
        rpath = cfg.train.resume_path or model_file  # "" -> this run's own model file
        got = checkpoint.load_model_dict(rpath, mdl, opt, load_opt=cfg.train.load_opt,
                                         strict=cfg.train.strict_load, arena=arena)
        if got is not None:
            num_it = got["num_it"] or 0
            if rank == 0:
                print(f"[{cfg.uid}] resumed {rpath} at iteration {num_it}")
        elif rank == 0:
            print(f"[{cfg.uid}] no existing model in {rpath}, starting from scratch")
    bs = max(cfg.train.bs // world, 1)
    n_ev = cfg.ds.vsitu.num_ev
    nb = 1 if cfg.overfit_batch else 2
    if cfg.task_type == "vb_arg":  # SRL rows: token sequences + pre-extracted features
        from vidsitu_amd.mdl_sf_base import get_head_dim

        batches = [synth_data.synth_srl_batch(comm, bs, n_ev, feat_dim=get_head_dim(cfg),
                                              seed=cfg.synth.seed + 17 * i + rank, device=dev) for i in range(nb)]
    else:
        batches = [synth_data.synth_batch(cfg, comm, bs, n_ev, seed=cfg.synth.seed + 17 * i + rank, device=dev)
                   for i in range(nb)]
    if not (cfg.only_val or cfg.only_test):
        mdl.train()
        t0, losses = time.time(), []
        for it in range(steps):
            b = batches[it % len(batches)]
            opt.zero_grad()
            loss = loss_fn(mdl(b), b)["loss"]
            loss.backward()
            opt.step(world=arena.all_reduce())
            losses.append(loss.detach())
        torch.cuda.synchronize()
        if rank == 0:
            ls = [round(float(x), 4) for x in losses]
            print(f"[{cfg.uid}] {steps} steps, {bs * n_ev * world * steps / (time.time() - t0):.1f} clips/s, "
                  f"loss {ls[0]} -> {ls[-1]}")
            checkpoint.save_model_dict(model_file, mdl, opt, num_it=num_it + steps, cfg=None)
            print(f"[{cfg.uid}] saved {model_file}")
    loss_d, acc_d = eval_fn(mdl, loss_fn, batches, "valid", rank)
    if rank == 0:
        print(f"[{cfg.uid}] valid {loss_d} {acc_d}")
    if cfg.do_dist:
        dist.destroy_process_group()


def main_dist(uid, **kwargs):
    steps = int(kwargs.pop("steps", 10))
    cfg = get_cfg(kwargs)
    cfg.uid = uid
    assert torch.cuda.is_available(), "the HIP path needs a GPU (no CPU fallback)"
    n = torch.cuda.device_count()
    cfg.num_gpus = n if kwargs.get("num_gpus") is None else int(kwargs["num_gpus"])
    cfg.do_dist = cfg.num_gpus > 1
    cfg.freeze()
    if cfg.do_dist:
        port = int(os.environ.get("MASTER_PORT", 0)) or _free_port()
        mp.spawn(main_fn, args=(cfg, steps, port), nprocs=cfg.num_gpus, join=True)
    else:
        main_fn(0, cfg, steps)


if __name__ == "__main__":
    mp.set_start_method("spawn", force=True)
    if len(sys.argv) < 2:
        sys.exit("usage: main_dist.py <uid> [--dotted.key=value ...] [--steps=N]")
    kw = {}
    for a in sys.argv[2:]:
        assert a.startswith("--") and "=" in a, f"bad argument {a}"
        k, v = a[2:].split("=", 1)
        kw[k] = v
    main_dist(sys.argv[1], **kw)
