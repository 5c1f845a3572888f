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
timings of every C-ABI entry point, taken in a separate instrumented pass on the launch stream
(algorithmic FLOPs or bytes of each launch / its measured duration); `cpu_baseline` times the fp32 torch oracle restatement
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
                    choices=["feat_fwd", "sf_txenc_train", "srl_gen_gpt2", "srl_gen_txdec"])
    ap.add_argument("--graph", type=int, default=1, help="replay the step from a hipGraph")
    ap.add_argument("--overlap", type=int, default=-1,
                    help="train: split the step into segment graphs and all-reduce finished gradient "
                         "buckets behind the remaining backward (default: on when WORLD_SIZE > 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


def _v(a):
    """ctypes scalar / plain python number -> python number."""
    return a.value if hasattr(a, "value") else a


def _nn(a):
    """1 if a pointer argument is non-null."""
    return 0 if a is None or _v(a) in (None, 0) else 1


_WAVES = {(128, 128): (2, 2), (64, 128): (1, 4), (128, 64): (2, 2), (64, 64): (2, 2),
          (256, 32): (4, 1), (256, 16): (4, 1), (256, 128): (4, 1), (128, 256): (1, 4)}


class EntryProbe:
    """Brackets EVERY C-ABI entry point (vidsitu_amd._lib.call) with HIP events recorded on the
    launch stream and aggregates per kernel family.  The label of a conv launch is the kernel
    template instance the library's own plan (vs_conv_plan) selects, spelled as rocprofv3 prints
    it, so the families line up with profiles/*kernel_stats*.csv.  While the step is enqueued
    the GPU is parked behind a spin kernel, so the launches run back to back and an event pair
    measures kernel time, not the Python launch latency in front of it."""

    def __init__(self):
        self.records = []

    def install(self):
        from vidsitu_amd import _lib

        self._lib = _lib
        lib = _lib.load()
        probe = self
        import ctypes as C

        def conv_label(d, dgrad, bnb=False):
            out = (C.c_int * 5)()
            lib.vs_conv_plan(C.byref(d), dgrad, out)
            bm, bn, ring, S, direct = list(out)
            taps = d.kT * d.kH * d.kW
            unit = d.sT == 1 and d.sH == 1 and d.sW == 1
            pw = taps == 1 and d.pT == 0 and d.pH == 0 and d.pW == 0
            mode = (0 if (pw and unit) else (1 if unit else 2)) if dgrad else (0 if pw else 1)
            if direct:
                ncols = d.Cin if dgrad else d.Cout
                K = taps * (d.Cout if dgrad else d.Cin)
                return f"conv_direct_kernel<{1 if ncols <= 16 else 2}, {min((K + 31) // 32, 5)}, {mode}>"
            wm, wn = _WAVES[(bm, bn)]
            fast = "true" if taps <= 31 else "false"
            tail = f", 0, {ring}, {'true' if bnb else 'false'}" if taps <= 31 else ""
            name = f"conv_igemm_kernel<{bm}, {bn}, {wm}, {wn}, {mode}, {fast}{tail}>"
            return name + (f" +splitk{S}" if S > 1 else "")

        def describe(name, a):
            """-> (family label, bound, algorithmic flops, algorithmic bytes)"""
            if name in ("vs_conv_fwd", "vs_conv_dgrad", "vs_conv_wgrad", "vs_conv_dgrad_bnstats"):
                d = a[3]._obj
                taps = d.kT * d.kH * d.kW
                mo = d.N * d.To * d.Ho * d.Wo
                mi = d.N * d.Ti * d.Hi * d.Wi
                flops = 2.0 * mo * d.Cout * d.Cin * taps
                byts = 2.0 * (mi * d.Cin + mo * d.Cout + d.Cout * d.Cin * taps)
                if name == "vs_conv_wgrad":
                    return "conv_wgrad (conv_wgrad_ring_kernel | conv_wgrad_kernel, + wgrad_reduce_kernel)", "mfma", flops, byts
                if name == "vs_conv_fwd" and (d.flags & 2):
                    byts += 2.0 * mo * d.Cout
                if name == "vs_conv_dgrad_bnstats":  # + the producer's saved conv output, read by the epilogue
                    return conv_label(d, 1, True), "mfma", flops, byts + 2.0 * mi * d.Cin
                return conv_label(d, 1 if name == "vs_conv_dgrad" else 0), "mfma", flops, byts
            if name in ("vs_stem_conv_fwd", "vs_stem_conv_wgrad"):
                n, t, h, w, cout, kt = [_v(x) for x in a[3:9]]
                ho, wo = (h + 6 - 7) // 2 + 1, (w + 6 - 7) // 2 + 1
                flops = 2.0 * n * t * ho * wo * cout * 3 * kt * 49  # the 3 real input channels
                byts = 2.0 * (n * t * h * w * 4 + n * t * ho * wo * cout)
                return ("stem_conv_kernel" if name == "vs_stem_conv_fwd" else
                        "stem_wgrad_kernel (+stem_slab_reduce_kernel)"), "mfma", flops, byts
            if name == "vs_bn_apply":
                rows, c = _v(a[5]), _v(a[6])
                return "bn_apply_cols_kernel", "hbm", 0.0, 2.0 * rows * c * (2 + _nn(a[3]))
            if name == "vs_bn_bwd_reduce":
                rows, c = _v(a[8]), _v(a[9])
                return "bn_bwd_reduce_kernel", "hbm", 0.0, 2.0 * rows * c * (2 + _nn(a[1]))
            if name == "vs_bn_bwd_apply":
                rows, c = _v(a[11]), _v(a[12])
                n_t = 2 + _nn(a[1]) + 1 + _nn(a[10])  # dz, y (+z) in; dy (+dres) out
                return "bn_bwd_apply_cols_kernel", "hbm", 0.0, 2.0 * rows * c * n_t
            if name == "vs_adam_step_dev" or name == "vs_adam_step":
                return "adam_kernel", "hbm", 0.0, 4.0 * _v(a[4]) * 7  # p,g,m,v in; p,m,v out
            return name.replace("vs_", "") + " (entry point)", None, 0.0, 0.0

        def hook(name, args, fn):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*args)
            e1.record()
            probe.records.append((describe(name, args), e0, e1))
            return rc

        _lib._probe = hook

    def remove(self):
        self._lib._probe = None

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for (name, bound, flops, byts), e0, e1 in self.records:
            a = agg.setdefault(name, [0, 0.0, 0.0, 0.0, bound])
            a[0] += 1
            a[1] += e0.elapsed_time(e1)
            a[2] += flops
            a[3] += byts
        return agg


def park_gpu(ms):
    """Keep the GPU busy for ~ms with a spin kernel so that the launches enqueued behind it run
    back to back (profiling pass only)."""
    if not hasattr(park_gpu, "per_mcycle"):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(1000000)
        torch.cuda.synchronize()
        e0.record()
        torch.cuda._sleep(1000000)
        e1.record()
        torch.cuda.synchronize()
        park_gpu.per_mcycle = max(e0.elapsed_time(e1), 1e-3)
    torch.cuda._sleep(int(ms / park_gpu.per_mcycle * 1e6))


def load_pmc_traffic():
    """HBM bytes per launch from the separate rocprofv3 --pmc passes (tools/pmc_traffic.sh ->
    profiles/pmc_traffic.json; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f)
    except OSError:
        return {}


def bench_srl_gen(args, rank, world, dev):
    """BASELINE configs[4] (informational, not the headline metric): features -> TxEncoder -> SRL caption per
    event by beam search (beam 5, 60 tokens, min_len = max_len - 1 so every hypothesis runs the full length),
    GPT-2 medium (`srl_gen_gpt2`) or the fairseq-style 3-layer decoder (`srl_gen_txdec`, the reference's
    default).  8 event clips per GPU; ranks are replicas (no collective).  One step = one generation of the
    batch; value = event captions (one per clip) per second."""
    from vidsitu_amd import synth_data
    from vidsitu_amd.extended_config import get_cfg
    from vidsitu_amd.mdl_selector import get_mdl_loss_eval

    dec = "gpt2" if args.workload == "srl_gen_gpt2" else "txdec"
    max_len = 60
    cfg = get_cfg({"task_type": "vb_arg", "mdl.mdl_name": "sfpret_txe_txd_vbarg", "mdl.tx_dec_type": dec,
                   "gen.beam_size": 5, "gen.max_len_b": max_len, "gen.min_len": max_len - 1})
    comm = synth_data.make_comm(cfg)
    sel = get_mdl_loss_eval(cfg)
    torch.manual_seed(0)
    mdl = sel["mdl"](cfg=cfg, comm=comm).to(dev).eval()
    batch = synth_data.synth_srl_batch(comm, bs=CLIPS_PER_GPU // 4, n_ev=4, seq_len=60, seed=1234 + rank, device=dev)
    evl = sel["evl"](cfg, comm, dev)
    for _ in range(max(args.warmup, 2)):  # eager warm-up, then the per-step hipGraph capture
        evl.forward_one_batch(mdl, batch)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = evl.forward_one_batch(mdl, batch)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ntok = sum(len(v["tokens"]) for r in out for v in r["vb_output"].values())
    if rank == 0:
        print(json.dumps({
            "metric": f"clips/s SRL caption generation, beam 5 x {max_len} tokens, {dec} decoder",
            "value": round(CLIPS_PER_GPU * world * args.steps / dt, 2), "unit": "clips/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[4] shape: pre-extracted features -> TxEncoder -> {dec} "
                                   f"decoder, beam 5, {max_len} tokens, 8 event clips/GPU as 2 videos x 4 events",
                       "clips_per_gpu": CLIPS_PER_GPU, "hipgraph": True, "tokens_per_generation": ntok,
                       "device_side_search": True},
            "roofline": None, "cpu_baseline": None}))
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


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
    # VS_BENCH_FORCE_DIST=1: initialise RCCL even for one rank (exercises the distributed step on
    # a single-GPU box)
    if world > 1 or os.environ.get("VS_BENCH_FORCE_DIST") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)

    from vidsitu_amd import synth_data
    from vidsitu_amd.extended_config import get_cfg
    from vidsitu_amd.mdl_selector import get_mdl_loss_eval
    from vidsitu_amd.optim import ArenaAdam, ParamArena

    if args.workload.startswith("srl_gen"):
        return bench_srl_gen(args, rank, world, dev)
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

    dist_on = world > 1 or (dist.is_available() and dist.is_initialized())
    gate = {"on": dist_on}  # the rank-0-only instrumented pass must not enter a collective
    overlap = train and (args.overlap == 1 or (args.overlap < 0 and world > 1))
    segments = None
    if train:
        mdl.train()
        arena = ParamArena(mdl)
        arena.broadcast_params(0)
        opt = ArenaAdam(arena, lr=cfg.train.lr, betas=(0.9, 0.99))

        def fwd_bwd():
            arena.transposes_async()  # dgrad weight images of the last update, beside the forward
            opt.zero_grad()  # (every gradient is overwritten anyway; the memset overlaps the stems)
            out = mdl(batch)
            loss = loss_fn(out, batch)["loss"]
            loss.backward()
            arena._join_transposes()  # no-op unless no dgrad ran (keeps a captured graph closed)
            return loss

        if overlap:
            # Segment s: [python callable, gradient bucket that is complete after it].  The trunk's
            # manual backward is deferred out of autograd and run stage group by stage group; the
            # bucket of a finished group is all-reduced (RCCL, async on its own stream) while the
            # next group computes.  Arena order = registration order: s1..s3 | s4 | s5 | heads+TxEnc.
            trunk = mdl.sf_mdl
            trunk.defer_backward = True
            segs = list(trunk.BWD_SEGMENTS)
            ranges = arena.bucket_ranges([trunk.backward_segment_modules(sg) for sg in segs])
            segments = [(fwd_bwd, ranges[-1])]  # everything outside the trunk is done after autograd
            for sg, rg in zip(segs, ranges[:-1]):
                segments.append((lambda sg=sg: trunk.run_backward_segment(sg), rg))

            def step():
                works = []
                out = None
                for fn, (lo, hi) in segments:
                    r = fn()
                    out = r if out is None else out
                    if gate["on"]:
                        works.append(arena.all_reduce_range(lo, hi, async_op=True))
                for w in works:
                    if w is not None:
                        w.wait()
                opt.step(world=world, defer_transposes=True)
                return out
        else:
            def step():
                loss = fwd_bwd()
                if gate["on"]:
                    arena.all_reduce()
                opt.step(world=world, defer_transposes=True)
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
    seg_graphs = None
    if args.graph and segments is not None:
        # one hipGraph per segment + one for Adam; the RCCL calls stay outside the graphs
        try:
            seg_graphs, pool = [], None
            for fn, _ in segments + [(lambda: opt.step(world=world, defer_transposes=True), None)]:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool):
                    fn()
                pool = g.pool()
                seg_graphs.append(g)
            torch.cuda.synchronize()
            used_graph = True
        except Exception as e:
            if rank == 0:
                print(f"[bench] segmented hipGraph capture failed, running eagerly: {e!r}", file=sys.stderr)
            seg_graphs = None
            mdl.sf_mdl._deferred = None
            torch.cuda.synchronize()
    elif args.graph and not (train and dist_on):
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

    def run_segmented():
        works = []
        for g, (_, (lo, hi)) in zip(seg_graphs[:-1], segments):
            g.replay()
            if dist_on:
                works.append(arena.all_reduce_range(lo, hi, async_op=True))
        for w in works:
            if w is not None:
                w.wait()
        seg_graphs[-1].replay()

    def run_once():
        if seg_graphs is not None:
            run_segmented()
        elif graph is not None:
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
        gate["on"] = False
        # per-kernel durations are taken with every launch on ONE stream (the timed region runs the
        # two pathways and the weight gradients on parallel streams, where launches overlap and a
        # launch's wall duration no longer measures the kernel)
        from vidsitu_amd import trunk as _trunk
        saved_modes = (_trunk.VideoTrunk.dual_stream, _trunk._WgradLanes.enabled)
        _trunk.VideoTrunk.dual_stream, _trunk._WgradLanes.enabled = False, False
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        eager_ms = (time.perf_counter() - t0) * 1e3  # CPU time to enqueue one eager step
        gate["on"] = False
        probe = EntryProbe()
        probe.install()
        reps = 3
        try:
            for _ in range(reps):
                try:
                    park_gpu(1.3 * eager_ms + 10.0)
                except Exception:  # no spin kernel: event pairs then include launch gaps
                    pass
                step()
                torch.cuda.synchronize()
        finally:
            probe.remove()
            _trunk.VideoTrunk.dual_stream, _trunk._WgradLanes.enabled = saved_modes
        agg = probe.summary()
        pmc = load_pmc_traffic()

        def fam(name, v):
            n, ms, fl, by, bound = v
            out = {"kernel": name, "launches_per_step": n // reps,
                   "ms_per_step": round(ms / reps, 3), "avg_launch_us": round(ms / n * 1e3, 2)}
            if bound == "mfma" and by > 0:
                # which roof binds a GEMM-shaped family is its arithmetic intensity against the ridge
                # (2500 TFLOP/s / 8 TB/s = 312 flop/B): the 1x1x1 convolutions of the early stages and
                # most weight gradients sit below it -- their roof is HBM, not the matrix cores
                inten = fl / by
                out["flop_per_byte"] = round(inten, 1)
                out["mfma_frac"] = round(fl / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4)
                out["hbm_frac"] = round(by / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)
                if inten < PEAK_BF16_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9):
                    bound = "hbm"
            out["bound"] = bound
            if bound == "mfma":
                ach = fl / (ms * 1e-3) / 1e12
                out.update({"achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(ach / PEAK_BF16_TFLOPS, 4)})
            elif bound == "hbm":
                ach = by / (ms * 1e-3) / 1e9
                out.update({"achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                            "frac": round(ach / PEAK_HBM_GBS, 4)})
            out["algorithmic_bytes_per_launch"] = round(by / n)
            # PMC table rows whose kernel name starts with this family's name (a family such as
            # bn_bwd_apply_kernel covers several template instances): launch-weighted mean
            key = name.split(" (")[0].split(" +")[0]
            rows = [t for k, t in pmc.items() if k.startswith(key) and not k.startswith("_")]
            nl = sum(t["launches"] for t in rows)
            out["traffic"] = round(sum((t["fetch_bytes"] + t["write_bytes"]) * t["launches"]
                                       for t in rows) / nl) if nl else None
            return out

        fams = sorted(agg.items(), key=lambda kv: -kv[1][1])
        tot_ms = sum(v[1] for _, v in fams) / reps
        top = next((fam(k, v) for k, v in fams if v[4] is not None), None)
        roof = dict(top)
        roof["probed_ms_per_step"] = round(tot_ms, 3)
        roof["note"] = ("dominant entry point of the step by summed HIP-event time; separate "
                        "instrumented eager pass on one stream (the timed region replays a hipGraph "
                        "whose pathway / wgrad branches run concurrently)")
        conv = [v for k, v in fams if v[4] == "mfma"]
        cms, cfl, cby = sum(v[1] for v in conv), sum(v[2] for v in conv), sum(v[3] for v in conv)
        roof["all_conv"] = {"ms_per_step": round(cms / reps, 3),
                            "achieved_tflops": round(cfl / (cms * 1e-3) / 1e12, 2),
                            "algorithmic_gbs": round(cby / (cms * 1e-3) / 1e9, 1),
                            "flop_per_byte": round(cfl / max(cby, 1), 1)}
        roof["families"] = [fam(k, v) for k, v in fams[:16]]

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
                       "grad_allreduce": ("bucketed, overlapped with backward (4 segment graphs)"
                                          if segments is not None else
                                          ("single" if train else None)),
                       "model_tflops": round(value * flop_per_clip / 1e3, 2),
                       "frac_of_bf16_mfma_peak": round(value * flop_per_clip / 1e3 / world /
                                                       PEAK_BF16_TFLOPS, 4)},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if dist.is_available() and dist.is_initialized():
        dist.barrier()  # rank 0 is still in its instrumented pass / CPU baseline: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
