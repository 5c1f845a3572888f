#!/usr/bin/env python3
"""`python main_dist.py <uid> --dotted.key=value ...` -- the reference's entry point
(`main_dist.py:132-172`: cfg -> `launch_job` -> `main_fn` -> `learner_init` -> `Learner.fit`)
for the hot path, on synthetic batches (the 50 GB frame dataset is out of scope).

Kept: uid + dotted overrides, `num_gpus` from the visible devices, one process per GPU with a
localhost TCP rendezvous and backend `cfg.DIST_BACKEND` ("nccl" == RCCL on ROCm;
`utils/trn_dist_utils.py:5-42`), `get_mdl_loss_eval` plugin lookup, Adam(betas=(0.9, 0.99)) at
`train.lr`, per-rank batch = `train.bs // num_gpus` (`utils/dat_utils.py:42-43`).
`train.resume` / `train.resume_path` / `train.load_opt` / `train.strict_load` restore a checkpoint in the
reference's file format before training and `misc.tmp_path/models/<uid>.pth` is written after it
(`vidsitu_amd/checkpoint.py`; `utils/trn_utils.py:631-716`).

The training loop IS the measured step (`vidsitu_amd/train_step.py::TrainStep`, the one bench.py times):
`Learner.train_epoch`'s zero_grad -> forward -> loss -> backward -> DDP all-reduce -> optimizer.step
(`utils/trn_utils.py:590-615`, `main_dist.py:68-79`) runs once eagerly (allocator / stream warm-up), is
captured into hipGraphs (one graph per backward segment when a process group exists, so that each finished
gradient bucket is all-reduced over RCCL behind the remaining backward; bf16 bucket payload for more than
one rank) and replayed for the remaining iterations -- the next batch is copied into the graph's static
input tensors between replays.  `--graph=0` keeps the loop eager (same TrainStep, no capture).

Ranks: the reference spawns them from a parent that has touched CUDA (`mp.spawn`,
`utils/trn_dist_utils.py:34-39`); here the parent only COUNTS devices and starts one fresh interpreter per
GPU (`vidsitu_amd/dist_launch.py`), rendezvous on 127.0.0.1 at a free port (the reference hard-codes 9997,
`utils/trn_dist_utils.py:30`, so two jobs on one node collide).  Under `torch.distributed.run` (RANK /
WORLD_SIZE already set) the process is a rank and spawns nothing.
`task_type=vb_arg` rows train on the SRL batch contract (`synth_data.synth_srl_batch`).  An empty
`train.resume_path` means `<tmp_path>/models/<uid>.pth`, as `Learner.load_model_dict` does
(`utils/trn_utils.py:643-646`).
Not kept: MLflow / progress bars / per-epoch checkpoint rotation (`utils/trn_utils.py`, out of scope).
The loop never syncs with the host inside a step (the reference does twice: `trn_utils.py:600,610`).
"""
import os
import sys
import time

import torch
import torch.distributed as dist

from vidsitu_amd import checkpoint, dist_launch, synth_data
from vidsitu_amd.extended_config import get_cfg
from vidsitu_amd.mdl_selector import get_mdl_loss_eval
from vidsitu_amd.optim import ArenaAdam, ParamArena
from vidsitu_amd.train_step import TrainStep


def _load_batch(static, new):
    """Next batch into the tensors the captured graphs read."""
    for k, v in static.items():
        if torch.is_tensor(v):
            v.copy_(new[k], non_blocking=True)


def main_fn(rank, cfg, steps, graph=True):
    world = cfg.num_gpus
    # the device index is the rank's place on ITS node (LOCAL_RANK, set by dist_launch.rank_env and by
    # torch.distributed.run); RANK only equals it on one node
    local = int(os.environ.get("LOCAL_RANK", rank)) if os.environ.get("WORLD_SIZE") not in (None, "") else rank
    if cfg.do_dist:
        torch.cuda.set_device(local)
        dist.init_process_group(backend=cfg.DIST_BACKEND, rank=rank, world_size=world,
                                device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
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
    elif cfg.mdl.load_sf_pretrained and cfg.task_type == "vb":  # Kinetics init of the trunk (utils/trn_utils.py:358-375)
        checkpoint.load_sf_pretrained(cfg, mdl, log=(lambda m: print(f"[{cfg.uid}] {m}")) if rank == 0 else (lambda m: None))
        arena.refresh()
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
        static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batches[0].items()}
        # the default TrainStep: full zero_grad every step, overlapped bf16 buckets when world > 1
        ts = TrainStep(mdl, loss_fn, arena, opt, static, world=world, use_dist=cfg.do_dist,
                       grad_bf16=cfg.do_dist and world > 1)
        losses = []
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        # the rate printed below is that of the steps AFTER the eager warm-up step and the capture (it == 0 and the
        # capture at it == 1 take seconds; a 10-step run would otherwise report mostly those)
        t0, timed_from = time.time(), 0
        with torch.cuda.stream(side):
            for it in range(steps):
                _load_batch(static, batches[it % len(batches)])
                if graph and it == 1:
                    ts.capture()  # raises on failure: never a silent eager loop under the graph's name
                    _load_batch(static, batches[it % len(batches)])  # (capture does not execute the step)
                if it == (1 if graph else min(1, steps - 1)) and steps > 1:
                    torch.cuda.synchronize()
                    t0, timed_from = time.time(), it
                ts.run()
                losses.append(ts.loss.clone())
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        t_run = time.time() - t0
        if rank == 0:
            ls = [round(float(x), 4) for x in losses]
            mode = "hipGraph replay" if ts.graphs is not None else "eager"
            print(f"[{cfg.uid}] {steps} steps ({mode}, {len(ts.segments)} segment(s), "
                  f"{'bf16' if ts.grad_bf16 else 'fp32'} gradient payload, world {world}), "
                  f"{bs * n_ev * world * (steps - timed_from) / t_run:.1f} clips/s over the last {steps - timed_from} "
                  f"step(s), loss {ls[0]} -> {ls[-1]}")
            checkpoint.save_model_dict(model_file, mdl, opt, num_it=num_it + steps, cfg=None)
            print(f"[{cfg.uid}] saved {model_file}")
    loss_d, acc_d = eval_fn(mdl, loss_fn, batches, "valid", rank)
    if rank == 0:
        print(f"[{cfg.uid}] valid {loss_d} {acc_d}")
    if cfg.do_dist:
        dist.barrier()
        dist.destroy_process_group()


def main_dist(uid, **kwargs):
    steps = int(kwargs.pop("steps", 10))
    graph = kwargs.pop("graph", None)
    cfg = get_cfg(kwargs)
    # hipGraph replay is the default for the verb-prediction rows (the measured step); the SRL rows (`vb_arg`) keep an
    # eager TrainStep unless --graph=1 asks for a capture (which raises if the model syncs with the host)
    graph = (cfg.task_type == "vb") if graph is None else str(graph) not in ("0", "false", "False")
    cfg.uid = uid
    n = dist_launch.visible_gpus()  # counts devices, does not initialise one
    assert n > 0, "the HIP path needs a GPU (no CPU fallback)"
    want = n if kwargs.get("num_gpus") is None else int(kwargs["num_gpus"])
    in_rank = os.environ.get("WORLD_SIZE") not in (None, "")
    if in_rank:  # started by a launcher (our own parent, or torch.distributed.run)
        world = int(os.environ["WORLD_SIZE"])
        if kwargs.get("num_gpus") is not None and world != want:
            sys.exit(f"main_dist.py: --num_gpus={want} but the launcher set WORLD_SIZE={world}")
        want = world
    # ranks that must find a GPU on THIS node: under an external launcher that is LOCAL_WORLD_SIZE (torch.distributed.run
    # sets it; a multi-node job has WORLD_SIZE > the node's GPUs), else the whole job
    local_want = int(os.environ.get("LOCAL_WORLD_SIZE", want)) if in_rank else want
    if local_want > n:
        sys.exit(f"main_dist.py: {local_want} ranks on this node but {n} GPU(s) visible")
    cfg.num_gpus = want
    cfg.do_dist = want > 1 or os.environ.get("VS_FORCE_DIST") == "1"
    cfg.freeze()
    if cfg.do_dist and not in_rank:
        # the parent never touches the GPU: one fresh interpreter per rank
        sys.exit(dist_launch.launch_ranks(want, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                          port=int(os.environ.get("MASTER_PORT", 0)) or None))
    main_fn(int(os.environ.get("RANK", "0")) if in_rank else 0, cfg, steps, graph)


if __name__ == "__main__":
    if len(sys.argv) < 2:
        sys.exit("usage: main_dist.py <uid> [--dotted.key=value ...] [--steps=N] [--graph=0|1]")
    kw = {}
    for a in sys.argv[2:]:
        assert a.startswith("--") and "=" in a, f"bad argument {a}"
        k, v = a[2:].split("=", 1)
        kw[k] = v
    main_dist(sys.argv[1], **kw)
