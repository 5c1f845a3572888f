#!/usr/bin/env python3
"""Headline benchmark: clips/s of the SlowFast-R50 (+TxEncoder) hot path on N MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload ...]

For N > 1 the driver launches one rank per GPU through torch.distributed.run (RCCL);
clips are sharded across ranks (8 per GPU, weak scaling), the only collective is the
gradient all-reduce of the training workload.  Rank 0 prints ONE JSON line.

Workloads (BASELINE.json `configs`):
  feat_fwd        configs[1]: SlowFast-R50 feature extractor, eval, 8 x 3x32x224x224 clips
  sf_txenc_train  configs[2]: SlowFast-R50 + 6-layer TxEncoder verb prediction, fwd+bwd+Adam

`roofline` is computed for the dominant kernel family of the step from per-launch HIP-event
timings taken in a separate instrumented pass on the launch stream (algorithmic FLOPs of each
conv launch / its measured duration); `cpu_baseline` times the fp32 torch oracle restatement
(kind "port": the reference's own CPU path cannot run) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0
GFLOP_PER_CLIP_FWD = 100.615  # 50.308 GMAC conv, 2 FLOP/MAC
CLIPS_PER_GPU = 8


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=os.environ.get("VS_BENCH_WORKLOAD", "sf_txenc_train"),
                    choices=["feat_fwd", "sf_txenc_train"])
    ap.add_argument("--graph", type=int, default=1, help="replay the step from a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


def pick_tile(M, ncols):
    """Python twin of pick_tile() in csrc/conv_igemm.hip (for kernel attribution only)."""
    if ncols >= 128:
        t = ((M + 127) // 128) * ((ncols + 127) // 128)
        return (128 if t >= 512 else 64, 128)
    if ncols >= 64:
        return (128 if (M + 127) // 128 >= 512 else 64, 64)
    return (256, 32) if ncols >= 32 else (256, 16)


class ConvProbe:
    """Wraps ops.conv_* with HIP events on the launch stream; aggregates per kernel family."""

    def __init__(self):
        self.records = []

    def install(self):
        from vidsitu_amd import ops

        self.ops = ops
        self.saved = (ops.conv_fwd, ops.conv_dgrad, ops.conv_wgrad)
        probe = self

        def timed(fn, key_fn):
            def inner(*a, **kw):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                out = fn(*a, **kw)
                e1.record()
                probe.records.append((key_fn(*a, **kw), e0, e1))
                return out
            return inner

        def k_fwd(x, w, k, s, p, **kw):
            ys = ops.conv_out_shape(x.shape, w.shape[0], k, s, p)
            M = ys[0] * ys[2] * ys[3] * ys[4]
            K = x.shape[1] * k[0] * k[1] * k[2]
            bm, bn = pick_tile(M, ys[1])
            pw = k == (1, 1, 1) and p == (0, 0, 0)
            # real (unpadded) input channels for the algorithmic count
            cin = 3 if x.shape[1] == 8 and k[1] == 7 else x.shape[1]
            flops = 2.0 * M * ys[1] * cin * k[0] * k[1] * k[2]
            byts = 2.0 * (x.shape[0] * x.shape[1] * x.shape[2] * x.shape[3] * x.shape[4] + M * ys[1])
            return (f"conv_igemm_kernel<{bm},{bn}> mode{0 if pw else 1} (fwd)", flops, byts)

        def k_dgrad(dy, wt, xs, k, s, p, **kw):
            M = xs[0] * xs[2] * xs[3] * xs[4]
            bm, bn = pick_tile(M, xs[1])
            flops = 2.0 * dy.shape[0] * dy.shape[2] * dy.shape[3] * dy.shape[4] * dy.shape[1] * \
                xs[1] * k[0] * k[1] * k[2]
            byts = 2.0 * (dy.numel() + M * xs[1])
            return (f"conv_igemm_kernel<{bm},{bn}> (dgrad)", flops, byts)

        def k_wgrad(dy, x, k, s, p, **kw):
            cin = 3 if x.shape[1] == 8 and k[1] == 7 else x.shape[1]
            flops = 2.0 * dy.shape[0] * dy.shape[2] * dy.shape[3] * dy.shape[4] * dy.shape[1] * \
                cin * k[0] * k[1] * k[2]
            byts = 2.0 * (dy.numel() + x.numel())
            return ("conv_wgrad_kernel (wgrad)", flops, byts)

        ops.conv_fwd = timed(self.saved[0], k_fwd)
        ops.conv_dgrad = timed(self.saved[1], k_dgrad)
        ops.conv_wgrad = timed(self.saved[2], k_wgrad)

    def remove(self):
        self.ops.conv_fwd, self.ops.conv_dgrad, self.ops.conv_wgrad = self.saved

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for (name, flops, byts), e0, e1 in self.records:
            ms = e0.elapsed_time(e1)
            a = agg.setdefault(name, [0, 0.0, 0.0, 0.0])
            a[0] += 1
            a[1] += ms
            a[2] += flops
            a[3] += byts
        return agg


def cpu_baseline(workload, n_vocab):
    """fp32 torch oracle on the host cores, ONE clip of the batch (bounded sample)."""
    from oracle.slowfast_ref import SFBaseRef, default_sf_cfg, slow_index

    torch.manual_seed(0)
    cfg = default_sf_cfg()
    mdl = SFBaseRef(cfg, n_vocab)
    g = torch.Generator().manual_seed(1234)
    fast = torch.randn(1, 3, 32, 224, 224, generator=g)
    slow = fast.index_select(2, slow_index(32, 4))
    cores = torch.get_num_threads()
    if workload == "feat_fwd":
        mdl.eval()
        with torch.no_grad():
            t0 = time.perf_counter()
            mdl.forward_feats([slow, fast])
            dt = time.perf_counter() - t0
        sample = "1 clip (fast 3x32x224x224 + slow 3x8x224x224), eval forward to [1,2304] features, 1 iteration"
    else:
        mdl.train()
        opt = torch.optim.Adam(mdl.parameters(), lr=1e-4, betas=(0.9, 0.99))
        t0 = time.perf_counter()
        loss = torch.nn.functional.cross_entropy(mdl([slow, fast]), torch.zeros(1, dtype=torch.long))
        loss.backward()
        opt.step()
        dt = time.perf_counter() - t0
        sample = "1 clip, SFBase fwd+bwd+Adam (batch-norm over that one clip), 1 iteration, no TxEncoder"
    return {"value": round(1.0 / dt, 4), "unit": "clips/s", "cores": cores, "kind": "port",
            "sample": sample}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)

    from vidsitu_amd import synth_data
    from vidsitu_amd.extended_config import get_cfg
    from vidsitu_amd.mdl_selector import get_mdl_loss_eval
    from vidsitu_amd.optim import ArenaAdam, ParamArena

    train = args.workload == "sf_txenc_train"
    overrides = {"mdl.mdl_name": "sf_base_txenc" if train else "sf_base"}
    if train:
        overrides.update({"tx_dec.encoder_layers": 6})  # dropout stays at the reference's 0.1
    cfg = get_cfg(overrides)
    comm = synth_data.make_comm(cfg)
    torch.manual_seed(0)
    sel = get_mdl_loss_eval(cfg)
    mdl = sel["mdl"](cfg=cfg, comm=comm).to(dev)
    loss_fn = sel["loss"](cfg, comm)
    # 8 clips per GPU as 2 videos x 4 events (the TxEncoder then attends over 4 event tokens;
    # the reference's literal 5 does not divide 8 -- SURVEY.md 0.10)
    batch = synth_data.synth_batch(cfg, comm, bs=CLIPS_PER_GPU // 4, n_ev=4, seed=1234 + rank,
                                   device=dev, dtype=torch.bfloat16)

    if train:
        mdl.train()
        arena = ParamArena(mdl)
        arena.broadcast_params(0)
        opt = ArenaAdam(arena, lr=cfg.train.lr, betas=(0.9, 0.99))

        def step():
            opt.zero_grad()
            out = mdl(batch)
            loss = loss_fn(out, batch)["loss"]
            loss.backward()
            w = arena.all_reduce()
            opt.step(world=w)
            return loss
    else:
        mdl.eval()

        def step():
            with torch.no_grad():
                feats = mdl.forward_encoder(batch)
                return mdl.head(feats)

    # ---- warm-up (also the hipGraph capture warm-up) -----------------------------------
    graph, used_graph = None, False
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(max(args.warmup, 1)):
            out = step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    if args.graph and not (train and world > 1):
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = step()
            graph.replay()
            torch.cuda.synchronize()
            used_graph = True
        except Exception as e:  # capture is an optimisation, never a correctness path
            if rank == 0:
                print(f"[bench] hipGraph capture failed, running eagerly: {e!r}", file=sys.stderr)
            graph = None
            torch.cuda.synchronize()

    def run_once():
        if graph is not None:
            graph.replay()
        else:
            step()

    # ---- timed region: exactly K steps, barrier + sync on both sides --------------------
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_once()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    clips = CLIPS_PER_GPU * world * args.steps
    value = clips / dt
    flop_per_clip = GFLOP_PER_CLIP_FWD * (3.0 if train else 1.0)

    roof = None
    if rank == 0 and not args.no_roofline:
        probe = ConvProbe()
        probe.install()
        try:
            for _ in range(3):
                step()
        finally:
            probe.remove()
        agg = probe.summary()
        tot_ms = sum(a[1] for a in agg.values())
        tot_fl = sum(a[2] for a in agg.values())
        name, a = max(agg.items(), key=lambda kv: kv[1][1])
        ach = a[2] / (a[1] * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": name, "launches_per_step": a[0] // 3,
                "avg_launch_us": round(a[1] / a[0] * 1e3, 2), "achieved": round(ach, 2),
                "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4),
                "traffic": None,
                "all_conv": {"ms_per_step": round(tot_ms / 3, 3),
                             "achieved_tflops": round(tot_fl / (tot_ms * 1e-3) / 1e12, 2),
                             "algorithmic_gbs": round(sum(x[3] for x in agg.values()) /
                                                      (tot_ms * 1e-3) / 1e9, 1)},
                "families": {k: {"launches": v[0] // 3, "ms_per_step": round(v[1] / 3, 3),
                                 "tflops": round(v[2] / (v[1] * 1e-3) / 1e12, 2)}
                             for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}}

    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.workload, len(comm.vb_id_vocab))

    if rank == 0:
        line = {
            "metric": "clips/s (10s@32x224x224) SlowFast+TxEnc fwd+bwd" if train
            else "clips/s (10s@32x224x224) SlowFast-R50 feature extractor fwd",
            "value": round(value, 2), "unit": "clips/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": ("BASELINE configs[2]: SlowFast-R50 + 6-layer TxEnc verb-pred, "
                                    "fwd+bwd+Adam, dropout 0.1, 8 clips/GPU as 2 videos x 4 events" if train else
                                    "BASELINE configs[1]: SlowFast-R50 feature extractor only, eval, "
                                    "8 clips x 3x32x224x224 per GPU"),
                       "clips_per_gpu": CLIPS_PER_GPU, "hipgraph": used_graph,
                       "model_tflops": round(value * flop_per_clip / 1e3, 2),
                       "frac_of_bf16_mfma_peak": round(value * flop_per_clip / 1e3 / world /
                                                       PEAK_BF16_TFLOPS, 4)},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
