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
Beside the headline value the N = 1 train line carries two more measured legs (rank 0, outside the timed region):
`feat_fwd` = configs[1] (eval forward, BN shifts calibrated for the bf16 weight rounding on two other clips: the
arithmetic whose logits sit within 1e-3 of the fp32 oracle, `config.parity`) and `canonical_b8x5` = the reference's own
batch, 8 videos x 5 events = 40 clips per step.
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

_JSON_FD = None  # set in a rank of a distributed job: the real stdout (see main)


def emit(line):
    """The bench line: the only thing this process writes to its real stdout."""
    text = json.dumps(line) + "\n"
    if _JSON_FD is None:
        sys.stdout.write(text)
        sys.stdout.flush()
    else:
        os.write(_JSON_FD, text.encode())


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
    ap.add_argument("--grad-dtype", default="auto", choices=["auto", "fp32", "bf16"],
                    help="payload of the gradient all-reduce (auto: bf16 when WORLD_SIZE > 1, fp32 master "
                         "parameters / moments either way)")
    ap.add_argument("--clips-per-gpu", type=int, default=8,
                    help="PROBE ONLY: clips per GPU other than the BASELINE config's 8 (multiple of 4); the line is "
                         "marked config.probe and is not the headline metric")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-feat-fwd", action="store_true",
                    help="train workload: skip the feature-extractor forward leg (`feat_fwd` object of the line)")
    return ap.parse_args()


def _v(a):
    """ctypes scalar / plain python number -> python number."""
    return a.value if hasattr(a, "value") else a


def _nn(a):
    """1 if a pointer argument is non-null."""
    return 0 if a is None or _v(a) in (None, 0) else 1


# the weight-gradient family of the roofline (entry points vs_conv_wgrad and vs_conv_wgrad_group; the kernels named after
# "+" run behind those calls and their PMC bytes are charged to the family)
WGRAD_FAMILY = ("conv_wgrad (conv_wgrad_deep_group_kernel | conv_wgrad_ring_kernel | conv_wgrad_deep_kernel | "
                "conv_wgrad_kernel, + wgrad_reduce)")
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

        def conv_label(d, dgrad, bnb=False, bnb2=False):
            out = (C.c_int * 5)()
            lib.vs_conv_plan(C.byref(d), dgrad, out)
            bm, bn, ring, S, direct = list(out)
            taps = d.kT * d.kH * d.kW
            unit = d.sT == 1 and d.sH == 1 and d.sW == 1
            pw = taps == 1 and d.pT == 0 and d.pH == 0 and d.pW == 0
            mode = (0 if (pw and unit) else (1 if unit else 2)) if dgrad else (0 if pw else 1)
            tf = lambda v: "true" if v else "false"
            if direct == 2:  # halo-image kernel (conv_halo.hip): wave tile in 16-row / 16-column units,
                # weight-ring depth (reported in the `ring` slot), unrolled taps (in the `split` slot)
                return f"conv_halo_kernel<{bm // 32}, {bn // 32}, {tf(bnb)}, {ring}, {S}>"
            if direct == 3:  # persistent pointwise kernel (conv_pw.hip): column tile, BN sums, ablation switch
                return f"conv_pw_kernel<{bn}, {tf(bnb)}, 0>"
            if direct:  # small-channel kernel: column tiles, 32-wide K steps, mode, 16-row tiles in flight, BN sums
                ncols = d.Cin if dgrad else d.Cout
                K = taps * (d.Cout if dgrad else d.Cin)
                nt, ks = (1 if ncols <= 16 else 2), min((K + 31) // 32, 6)
                return f"conv_direct_kernel<{nt}, {ks}, {mode}, {4 if nt * ks <= 6 else 2}, false>"
            wm, wn = _WAVES[(bm, bn)]
            fast = "true" if taps <= 31 else "false"
            tail = f", 0, {ring}, {tf(bnb)}, {tf(bnb2)}" if taps <= 31 else ""
            name = f"conv_igemm_kernel<{bm}, {bn}, {wm}, {wn}, {mode}, {fast}{tail}>"
            return name + (f" +splitk{S}" if S > 1 else "")

        def describe(name, a):
            """-> (family label, bound, algorithmic flops, algorithmic bytes)"""
            if name in ("vs_conv_fwd", "vs_conv_dgrad", "vs_conv_wgrad", "vs_conv_dgrad_bnstats", "vs_conv_dgrad_ex"):
                d = a[3]._obj
                taps = d.kT * d.kH * d.kW
                mo = d.N * d.To * d.Ho * d.Wo
                mi = d.N * d.Ti * d.Hi * d.Wi
                flops = 2.0 * mo * d.Cout * d.Cin * taps
                byts = 2.0 * (mi * d.Cin + mo * d.Cout + d.Cout * d.Cin * taps)
                if name == "vs_conv_wgrad":
                    return WGRAD_FAMILY, "mfma", flops, byts
                if name == "vs_conv_fwd" and (d.flags & 2):
                    byts += 2.0 * mo * d.Cout
                if name == "vs_conv_dgrad_bnstats":  # + the producer's saved conv output, read by the epilogue
                    return conv_label(d, 1, True), "mfma", flops, byts + 2.0 * mi * d.Cin
                if name == "vs_conv_dgrad_ex":
                    ep = a[4]._obj
                    bnb, bnb2 = bool(ep.stats_partial), bool(ep.stats_partial2)
                    extra = (2.0 if ep.residual else 0.0) + (2.0 if bnb else 0.0) + (2.0 if bnb2 else 0.0) + \
                        (0.125 if ep.residual_bits else 0.0)
                    return conv_label(d, 1, bnb, bnb2), "mfma", flops, byts + extra * mi * d.Cin
                return conv_label(d, 1 if name == "vs_conv_dgrad" else 0), "mfma", flops, byts
            if name == "vs_conv_wgrad_group":  # several weight gradients in one launch: the sum of their algorithmic work
                flops = byts = 0.0
                for i in range(_v(a[1])):
                    d = a[0][i].d
                    taps = d.kT * d.kH * d.kW
                    mo, mi = d.N * d.To * d.Ho * d.Wo, d.N * d.Ti * d.Hi * d.Wi
                    flops += 2.0 * mo * d.Cout * d.Cin * taps
                    byts += 2.0 * (mi * d.Cin + mo * d.Cout + d.Cout * d.Cin * taps)
                return WGRAD_FAMILY, "mfma", flops, byts
            if name == "vs_bn_apply2":  # y and the shortcut's raw output in, the block's output + its ReLU bits out
                rows, c = _v(a[8]), _v(a[9])
                return "bn_apply_cols_kernel", "hbm", 0.0, 2.0 * rows * c * 3 + rows * c / 8.0 * _nn(a[7])
            if name == "vs_bn_bwd_apply2":  # dz + bits, two y in; two dy out
                rows, c = _v(a[16]), _v(a[17])
                return "bn_bwd_apply2_cols_kernel", "hbm", 0.0, 2.0 * rows * c * 5 + rows * c / 8.0
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
            if name == "vs_bn_apply_mask":  # the same kernel, + the ReLU bit mask (1 bit per element)
                rows, c = _v(a[6]), _v(a[7])
                return "bn_apply_cols_kernel", "hbm", 0.0, 2.0 * rows * c * (2 + _nn(a[3])) + rows * c / 8.0
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


def load_pmc_traffic(workload="sf_txenc_train"):
    """HBM bytes per launch from the separate rocprofv3 --pmc passes (tools/pmc_traffic.sh ->
    profiles/pmc_traffic.json, profiles/pmc_traffic_feat_fwd.json for the forward workload; FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json" if workload == "sf_txenc_train"
                        else f"pmc_traffic_{workload}.json")
    try:
        with open(path) as f:
            return json.load(f)
    except OSError:
        return {}


def eval_parity_note(calibrated=False, train=False):
    """The eval path's distance to the fp32 reference at the logits, as LAST MEASURED by
    tests/test_gpu_parity_full.py (one 224^2 SlowFast-R50 clip against the fp32 oracle), read from
    profiles/parity_eval.json (the file that test writes, committed with the commit it was measured at), for the eval
    mode this process runs in.  north_star asks for 1e-3: the plain bf16 path sits at ~3e-3, all of it the rounding of
    the fp32 master weights to bf16; split bf16 weights (VS_EVAL_SPLIT_WEIGHTS=1, every convolution twice) meet it, and
    so does `calibrated` = the eval forward bench.py times: bf16 weights with the rounding's per-channel constants
    folded into the BN shifts (SFBase.calibrate_weight_rounding on two other clips; no cost per forward).
    None when the file is absent: the line then carries no parity figures rather than remembered ones."""
    from vidsitu_amd.trunk import ResBlock, _Unit

    mode = "split_bf16_weights" if _Unit.split_weights else ("fp32_residual_stream" if ResBlock.residual_fp32
                                                             else ("bf16_calibrated_shift" if calibrated else "bf16"))
    try:
        with open(os.path.join(ROOT, "profiles", "parity_eval.json")) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        return None
    note = {"eval_mode": mode, "logits_rel_err_vs_fp32_oracle": rec["logits_rel_err_vs_fp32_oracle"].get(mode),
            # the tolerance is RELATIVE: the largest logit difference over the largest |logit| of the oracle (the test's
            # head has max |logit| in the thousands, so "1e-3" is not an absolute bound)
            "metric": "max|logit - oracle logit| / max|oracle logit|",
            "north_star": rec.get("north_star", 1e-3),
            "same_bf16_weights_both_sides": rec.get("same_bf16_weights_both_sides"),
            "measured_at_commit": rec.get("commit"), "source": rec.get("source", "") + "; not re-measured in this run"}
    rob = rec.get("robustness")
    if rob and calibrated:
        # calibrated shifts over 8 evaluation clips and under a calibration / evaluation distribution shift
        note["calibrated_shift_over_clips"] = {k: {"max": v["max"], "median": v["median"]}
                                               for k, v in rob.get("cases", {}).items()}
    if train:
        # a TRAINING line times batch-statistic arithmetic: its own measured distance, beside the eval figure
        tm = rec.get("train_mode")
        note["train_mode"] = None if tm is None else {**tm, "metric": note["metric"]}
        note["note"] = ("this line times the TRAINING step: `train_mode` is the distance of that arithmetic "
                        "(batch-statistic BN, bf16 storage) from the fp32 oracle; the eval figure is the same weights' "
                        "inference path")
    return note


def feat_fwd_leg(dev, rank, replays=20):
    """BASELINE configs[1] (SlowFast-R50 feature extractor only, eval, 8 x 3x32x224x224 bf16): one hipGraph of the
    forward to the [8, 2304] features, `replays` replays between two synchronisations.  Same model construction, batch
    and step as `--workload feat_fwd`."""
    from vidsitu_amd import synth_data
    from vidsitu_amd.extended_config import get_cfg
    from vidsitu_amd.mdl_selector import get_mdl_loss_eval

    cfg = get_cfg({"mdl.mdl_name": "sf_base"})
    comm = synth_data.make_comm(cfg)
    torch.manual_seed(0)
    mdl = get_mdl_loss_eval(cfg)["mdl"](cfg=cfg, comm=comm).to(dev)
    mdl.eval()
    batch = synth_data.synth_batch(cfg, comm, bs=2, n_ev=4, seed=1234 + rank, device=dev, dtype=torch.bfloat16)
    calibrated = calibrate_eval(mdl, cfg, comm, dev, rank)

    def step():
        with torch.no_grad():
            return mdl.head(mdl.forward_encoder(batch))

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):  # the warm-up's stream: its cached scratch buffers are re-used, none is made in the graph
        step()
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(replays):
        g.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    cps = 8 * replays / dt
    return {"workload": "BASELINE configs[1]: SlowFast-R50 feature extractor only, eval, 8 clips x 3x32x224x224",
            "clips_per_s": round(cps, 1), "ms_per_step": round(dt / replays * 1e3, 3), "replays": replays,
            "hipgraph": True, "dtype": "bf16",
            "frac_of_bf16_mfma_peak": round(cps * GFLOP_PER_CLIP_FWD / 1e3 / PEAK_BF16_TFLOPS, 4),
            "parity": eval_parity_note(calibrated)}


def calibrate_eval(mdl, cfg, comm, dev, rank):
    """Before any timing: the eval model measures, on two clips OTHER than the timed batch (seed 999 + rank), the
    per-channel constants the bf16 rounding of its convolution weights adds, and folds their correction into the BN
    shifts (SFBase.calibrate_weight_rounding).  The timed forward runs the same launches on the same bytes.
    VS_EVAL_CALIBRATE=0: off (the plain bf16 folds)."""
    from vidsitu_amd import synth_data

    if os.environ.get("VS_EVAL_CALIBRATE", "1") == "0":
        return False
    cal = synth_data.synth_batch(cfg, comm, bs=1, n_ev=2, seed=999 + rank, device=dev, dtype=torch.bfloat16)
    mdl.calibrate_weight_rounding(cal)
    return True


def canonical_b8x5_leg(dev, rank, replays=10):
    """"also report the canonical [B, 5] shape at B = 8" -- the reference's own batch (train.bs = 8 videos
    x 5 events = 40 clips per step; configs/vsitu_cfg.yml, main_dist.py): the same model, optimizer and TrainStep as the
    headline line at [B = 8, E = 5], one whole-step hipGraph, `replays` replays between two synchronisations.  Rank 0,
    outside the timed region."""
    from vidsitu_amd import synth_data
    from vidsitu_amd.extended_config import get_cfg
    from vidsitu_amd.mdl_selector import get_mdl_loss_eval
    from vidsitu_amd.optim import ArenaAdam, ParamArena
    from vidsitu_amd.train_step import TrainStep

    cfg = get_cfg({"mdl.mdl_name": "sf_base_txenc", "tx_dec.encoder_layers": 6})
    comm = synth_data.make_comm(cfg)
    torch.manual_seed(0)
    sel = get_mdl_loss_eval(cfg)
    mdl = sel["mdl"](cfg=cfg, comm=comm).to(dev)
    mdl.train()
    batch = synth_data.synth_batch(cfg, comm, bs=8, n_ev=5, seed=1234 + rank, device=dev, dtype=torch.bfloat16)
    arena = ParamArena(mdl)
    opt = ArenaAdam(arena, lr=cfg.train.lr, betas=(0.9, 0.99))
    ts = TrainStep(mdl, sel["loss"](cfg, comm), arena, opt, batch, world=1, use_dist=False, grad_fill="learn")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            ts.step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    ts.capture()
    ts.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(replays):
        ts.run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    cps = 40 * replays / dt
    loss = float(ts.loss) if ts.loss is not None else None
    del ts, opt, arena, mdl
    torch.cuda.empty_cache()
    return {"workload": "the reference's own batch: [B = 8 videos, E = 5 events] = 40 clips per step, SlowFast-R50 + "
                        "6-layer TxEnc, fwd+bwd+Adam, dropout 0.1",
            "clips_per_s": round(cps, 1), "ms_per_step": round(dt / replays * 1e3, 3), "replays": replays,
            "hipgraph": True, "dtype": "bf16", "loss_finite": bool(loss is not None and loss == loss),
            "frac_of_bf16_mfma_peak": round(cps * GFLOP_PER_CLIP_FWD * 3.0 / 1e3 / PEAK_BF16_TFLOPS, 4)}


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
        emit(({
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


def _cpu_model_string():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _mem_available_gb():
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable"):
                    return int(line.split()[1]) / 1e6
    except OSError:
        pass
    return 0.0


def cpu_baseline(workload, n_vocab, budget_s=60.0):
    """SURVEY.md 8(d) / BASELINE.md 3: the fp32 torch restatement of the reference's path (the oracle;
    kind "port" -- the reference's own CPU path cannot execute, its arithmetic lives in un-vendored
    packages) on the host cores, same synthetic inputs as the GPU workload: N = 8 clips, 1 warm-up + 3
    timed iterations, the thread count calibrated (see below; all hardware threads is the slowest choice
    on a 256-thread host).
      feat_fwd        eval forward to the [8, 2304] features
      sf_txenc_train  SlowFast-R50 + vid_feat_encoder + 6-layer TxEncoder (the reference's per-head-loop
                      algorithm, oracle/txenc_ref.py) + Linear(1024, V): forward + backward + Adam,
                      batch statistics over the 8 clips (2 videos x 4 events)
    Bounded: the sample shrinks to 4 / 2 / 1 clips when the host has too little memory for 8 (fp32
    autograd keeps about 4 GB of activations per clip) or when the projected iteration time does not fit
    `budget_s`, and the timed iterations stop early once the budget is spent; `sample` states what was
    actually run."""
    from oracle import txenc_ref
    from oracle.slowfast_ref import SFBaseRef, default_sf_cfg, slow_index

    train = workload != "feat_fwd"
    torch.manual_seed(0)
    cfg = default_sf_cfg()
    mdl = SFBaseRef(cfg, n_vocab)
    g = torch.Generator().manual_seed(1234)
    # Thread count: torch's CPU convolutions do not scale to every hardware thread of a large host (on
    # the 256-thread GPU box one 8-clip training iteration took 365 s with 256 threads -- slower than 8
    # cores): time ONE clip's eval forward at a few thread counts (a few seconds in all) and keep the
    # fastest; `cores` reports the count actually used.
    ncpu = os.cpu_count() or 1
    probe = torch.randn(1, 3, 32, 224, 224, generator=g)
    probe = [probe.index_select(2, slow_index(32, 4)), probe]
    mdl.eval()
    best, t_cal = (None, 1e9), time.perf_counter()
    for th in sorted({min(ncpu, c) for c in (8, 16, 32, 64, 128)}):
        torch.set_num_threads(th)
        with torch.no_grad():
            if best[0] is None:
                mdl.forward_feats(probe)  # first touch
            t1 = time.perf_counter()
            mdl.forward_feats(probe)
            dt1 = time.perf_counter() - t1
        if dt1 < best[1]:
            best = (th, dt1)
        if time.perf_counter() - t_cal > 25.0:
            break
    threads, t_clip_fwd = best
    torch.set_num_threads(threads)
    n = 8
    need_gb = (4.0 if train else 1.0)  # measured: 3.8 GB peak RSS for one training clip
    avail = _mem_available_gb()
    per_iter = (lambda k: k * t_clip_fwd * (3.5 if train else 1.0))  # projected seconds per iteration
    while n > 1 and ((avail > 0 and n * need_gb + 8.0 > avail) or 2.2 * per_iter(n) > budget_s):
        n //= 2
    g = torch.Generator().manual_seed(1234)
    fast = torch.randn(n, 3, 32, 224, 224, generator=g)
    slow = fast.index_select(2, slow_index(32, 4))
    if not train:
        mdl.eval()

        def it():
            with torch.no_grad():
                mdl.forward_feats([slow, fast])
        what = f"{n} clips (fast 3x32x224x224 + slow 3x8x224x224), eval forward to [{n}, 2304] features"
    else:
        mdl.train()
        n_ev = 4 if n % 4 == 0 else n
        enc = torch.nn.Sequential(torch.nn.Linear(2304, 1024), torch.nn.ReLU(), torch.nn.Linear(1024, 1024))
        txw = {k: torch.nn.Parameter(torch.from_numpy(v)) for k, v in txenc_ref.make_weights(1024, 1024, 6, 0).items()}
        out = torch.nn.Linear(1024, n_vocab)
        params = list(mdl.sf_mdl.parameters()) + list(enc.parameters()) + list(txw.values()) + list(out.parameters())
        opt = torch.optim.Adam(params, lr=1e-4, betas=(0.9, 0.99))
        labels = torch.randint(0, n_vocab, (n,), generator=g)

        def it():
            opt.zero_grad()
            feats = mdl.forward_feats([slow, fast]).view(n // n_ev, n_ev, -1)
            x = enc(feats)
            for i in range(6):
                x = txenc_ref.encoder_layer(x, txw, f"layers.{i}.", 8)
            loss = torch.nn.functional.cross_entropy(out(x).view(n, -1), labels)
            loss.backward()
            opt.step()
        what = (f"{n} clips as {n // n_ev} videos x {n_ev} events, SlowFast-R50 + vid_feat_encoder + 6-layer "
                f"TxEncoder (per-head loop, dropout off) + Linear(1024, {n_vocab}): fwd + bwd + Adam, "
                f"batch-norm over the {n} clips")
    t0 = time.perf_counter()
    it()  # warm-up
    warm = time.perf_counter() - t0
    times = []
    for _ in range(3):
        spent = time.perf_counter() - t0
        if spent + (times[-1] if times else warm) > budget_s:
            break
        t1 = time.perf_counter()
        it()
        times.append(time.perf_counter() - t1)
    if times:
        dt = sorted(times)[len(times) // 2]
        how = f"1 warm-up ({warm:.1f} s) + {len(times)} timed iteration(s), median {dt:.2f} s"
    else:  # the warm-up alone used the budget: it is the sample
        dt = warm
        how = f"1 cold iteration of {warm:.1f} s (no budget left for timed iterations)"
    return {"value": round(n / dt, 4), "unit": "clips/s", "cores": threads, "kind": "port",
            "cpu_model": _cpu_model_string(),
            "sample": f"{what}; {how}; torch {torch.__version__} fp32, {threads} threads"}


def main():
    global CLIPS_PER_GPU
    args = parse()
    if args.clips_per_gpu != 8:
        assert args.clips_per_gpu % 4 == 0 and args.clips_per_gpu > 0, "--clips-per-gpu: a multiple of 4 (videos x 4 events)"
        CLIPS_PER_GPU = args.clips_per_gpu
    from vidsitu_amd import dist_launch

    # VS_BENCH_FORCE_DIST=1: initialise RCCL even for one rank (exercises the launcher and the distributed
    # step on a single-GPU box)
    force_dist = os.environ.get("VS_BENCH_FORCE_DIST") == "1"
    # TEST ONLY (tests/test_gpu_main_dist.py): every rank on cuda:0 and a gloo group, so that the N > 1 code path of
    # this file -- launcher, barriers, bucketed overlapped all-reduce, max-over-ranks timing, rank 0's instrumented pass
    # while the others wait -- runs on a one-GPU box.  The numbers of such a run mean nothing (the ranks share the chip).
    share_gpu = os.environ.get("VS_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("VS_BENCH_DIST_BACKEND", "nccl")
    try:
        world, spawn = dist_launch.world_from_env(args.gpus)
    except ValueError as e:  # WORLD_SIZE from a launcher disagrees with --gpus: never a silent N' != N line
        sys.exit(f"bench.py: {e}")
    if os.environ.get("WORLD_SIZE") in (None, "") and (spawn or force_dist):
        # `python bench.py --gpus N` without a launcher: THIS process becomes the launcher.  It has not
        # touched the GPU (importing torch does not) and never will: N fresh children, one per GPU, rank
        # environment + a free rendezvous port; rank 0's JSON line passes through; any failing rank fails
        # the job (the reference: `launch_job`, utils/trn_dist_utils.py:32-39).
        sys.exit(dist_launch.launch_ranks(world, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                          check_devices=not share_gpu))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    global _JSON_FD
    if world > 1 or force_dist:
        # RCCL prints to the process's stdout ("Librccl path : ..."): the contract is ONE JSON line there.  Keep a
        # private copy of the real stdout for that line and point fd 1 at stderr for everything else.
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            sys.exit("bench.py: WORLD_SIZE is set but MASTER_PORT is not (launch through torch.distributed.run "
                     "or plain `python bench.py --gpus N`)")
        if share_gpu:
            local_rank = 0
        if torch.cuda.device_count() <= local_rank:
            sys.exit(f"bench.py: rank {rank} wants cuda:{local_rank} but {torch.cuda.device_count()} GPU(s) are visible")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
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
        # dropout stays at the reference's 0.1.  VS_BENCH_ENC_LAYERS: probe only (cost of the 8-token section per layer)
        overrides.update({"tx_dec.encoder_layers": int(os.environ.get("VS_BENCH_ENC_LAYERS", "6"))})
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
    ts = None
    if train:
        from vidsitu_amd.train_step import TrainStep

        mdl.train()
        arena = ParamArena(mdl)
        arena.broadcast_params(0)
        opt = ArenaAdam(arena, lr=cfg.train.lr, betas=(0.9, 0.99))
        overlap = None if args.overlap < 0 else bool(args.overlap)
        grad_bf16 = dist_on and (args.grad_dtype == "bf16" or (args.grad_dtype == "auto" and world > 1))
        ts = TrainStep(mdl, loss_fn, arena, opt, batch, world=world, overlap=overlap, use_dist=dist_on,
                       grad_bf16=grad_bf16, grad_fill="learn")
        step = ts.step
    else:
        mdl.eval()
        calibrated_main = calibrate_eval(mdl, cfg, comm, dev, rank)

        def step():
            with torch.no_grad():
                feats = mdl.forward_encoder(batch)
                return mdl.head(feats)

    # ---- warm-up (also the hipGraph capture warm-up) -----------------------------------
    graph, used_graph = None, False
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    # the W untimed steps: eager ones first (allocator, plan caches, lane streams; two are enough), the rest as replays of
    # the captured graph -- a graph's FIRST launch uploads it (a one-time ~1 ms that is not part of a step)
    n_eager = max(1, min(args.warmup, 2)) if args.graph else max(args.warmup, 1)
    with torch.cuda.stream(side):
        for _ in range(n_eager):
            out = step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    if args.graph:
        # A failed capture is an error, not a fallback: an eager line under the same metric name
        # would be a different measurement (the process exits non-zero).
        if ts is not None:
            ts.capture()
        else:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):  # (the warm-up's stream: see TrainStep.capture)
                out = step()
        used_graph = True

    def run_once():
        if ts is not None:
            ts.run()
        elif graph is not None:
            graph.replay()
        else:
            step()

    if used_graph:
        for _ in range(max(args.warmup - n_eager, 1)):
            run_once()
        torch.cuda.synchronize()

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
        if ts is not None:
            ts.collectives = False  # the rank-0-only instrumented pass must not enter a collective
        # per-kernel durations are taken with every launch on ONE stream (the timed region runs the
        # two pathways and the weight gradients on parallel streams, where launches overlap and a
        # launch's wall duration no longer measures the kernel)
        from vidsitu_amd import trunk as _trunk
        from vidsitu_amd import _lib as _vslib
        # kernel launches of one step as the timed region runs it (default modes: two streams, pair launches);
        # the captured graph replays exactly this sequence
        n0 = _vslib.load().vs_launch_count()
        step()
        torch.cuda.synchronize()
        launches_per_step = int(_vslib.load().vs_launch_count() - n0)
        # one stream, no lanes, no (dgrad, wgrad) pair launches: an event pair then brackets exactly one entry point's kernels
        from vidsitu_amd import ops as _ops
        saved_modes = (_trunk.VideoTrunk.dual_stream, _trunk._WgradLanes.enabled, _trunk._Unit.pair_launch)
        _trunk.VideoTrunk.dual_stream, _trunk._WgradLanes.enabled, _trunk._Unit.pair_launch = False, False, False
        # ... and a weight gradient's slab reduce stays behind ITS entry point (in the timed region it shares the next
        # unit's BN-backward finalize launch: ops.REDUCE_MERGE), so the family it is charged to is the one it belongs to
        saved_merge, _ops.REDUCE_MERGE = _ops.REDUCE_MERGE, False
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        eager_ms = (time.perf_counter() - t0) * 1e3  # CPU time to enqueue one eager step
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
            _trunk.VideoTrunk.dual_stream, _trunk._WgradLanes.enabled, _trunk._Unit.pair_launch = saved_modes
            _ops.REDUCE_MERGE = saved_merge
        agg = probe.summary()
        pmc = load_pmc_traffic(args.workload)

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
            # PMC table rows of this family: the kernels whose name starts with the family's name are
            # launched once per entry-point call (a family such as bn_bwd_apply_kernel covers several
            # template instances): launch-weighted mean; kernels named after a "+" in the label
            # (wgrad_reduce_kernel behind conv_wgrad, the stem's slab reduce) run behind that call and
            # their bytes are charged to it
            key = name.split(" (")[0].split(" +")[0]
            rows = [t for k, t in pmc.items() if k.startswith(key) and not k.startswith("_")]
            extra_names = [e.strip(" )") for e in name.split("+")[1:]]
            extra = [t for k, t in pmc.items() if any(k.startswith(e) for e in extra_names if e)]
            nl = sum(t["launches"] for t in rows)
            out["traffic"] = round(sum((t["fetch_bytes"] + t["write_bytes"]) * t["launches"]
                                       for t in rows + extra) / nl) if nl else None
            return out

        fams = sorted(agg.items(), key=lambda kv: -kv[1][1])
        tot_ms = sum(v[1] for _, v in fams) / reps
        top = next((fam(k, v) for k, v in fams if v[4] is not None), None)
        roof = dict(top)
        roof["probed_ms_per_step"] = round(tot_ms, 3)
        roof["traffic_source"] = ("profiles/pmc_traffic" + ("" if train else "_" + args.workload) + ".json (" + str(pmc.get("_meta", {}).get("build", "build not recorded")) +
                                  "): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/pmc_traffic.sh, "
                                  "not collected in this run")
        roof["note"] = ("dominant entry point of the step by summed HIP-event time; separate "
                        "instrumented eager pass on one stream (the timed region replays a hipGraph "
                        "whose pathway / wgrad branches run concurrently)")
        conv = [v for k, v in fams if v[4] == "mfma"]
        cms, cfl, cby = sum(v[1] for v in conv), sum(v[2] for v in conv), sum(v[3] for v in conv)
        # library kernels of one step (a few torch-native fills / copies not counted); the family's own count stays in
        # `launches_per_step`
        roof["total_launches_per_step"] = launches_per_step
        roof["all_conv"] = {"ms_per_step": round(cms / reps, 3),
                            "achieved_tflops": round(cfl / (cms * 1e-3) / 1e12, 2),
                            "frac_of_bf16_mfma_peak": round(cfl / (cms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                            "algorithmic_gbs": round(cby / (cms * 1e-3) / 1e9, 1),
                            "flop_per_byte": round(cfl / max(cby, 1), 1)}
        # every batch-norm pass and its glue (finalize / partial reduces are entry points without a byte model)
        bn = [(k, v) for k, v in fams if k.startswith("bn_")]
        if bn:
            bms, bby, bn_l = sum(v[1] for _, v in bn), sum(v[3] for _, v in bn), sum(v[0] for _, v in bn)
            pmc_b = 0.0
            for k, v in bn:  # PMC bytes per launch (launch-weighted over the family's template instances) x launches
                key = k.split(" (")[0]
                rows = [t for kk, t in pmc.items() if kk.startswith(key) and not kk.startswith("_")]
                nl = sum(t["launches"] for t in rows)
                if nl:
                    pmc_b += sum((t["fetch_bytes"] + t["write_bytes"]) * t["launches"] for t in rows) / nl * v[0]
            roof["bn_all"] = {"ms_per_step": round(bms / reps, 3), "launches_per_step": bn_l // reps,
                              "algorithmic_gb_per_step": round(bby / reps / 1e9, 3),
                              "pmc_gb_per_step": round(pmc_b / reps / 1e9, 3) if pmc_b else None,
                              "algorithmic_gbs": round(bby / (bms * 1e-3) / 1e9, 1),
                              "tiny_launches_per_step": sum(v[0] for k, v in bn if v[4] is None) // reps}
        roof["families"] = [fam(k, v) for k, v in fams[:16]]

    # BASELINE configs[1] beside configs[2]: the feature extractor's eval forward (the `feat_fwd` workload) timed by the
    # same process AFTER the timed region, rank 0 only, 20 hipGraph replays -- so that the forward rate is a
    # driver-measured number too.  Not part of `value`.
    fwd = None
    canon = None
    if rank == 0 and train and args.graph and not args.no_feat_fwd:
        fwd = feat_fwd_leg(dev, rank)
        if world == 1 and CLIPS_PER_GPU == 8:
            canon = canonical_b8x5_leg(dev, rank)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:  # a reported baseline of the N = 1 line only
        cpu = cpu_baseline(args.workload, len(comm.vb_id_vocab))

    if rank == 0:
        line = {
            "metric": "clips/s (10s@32x224x224) SlowFast+TxEnc fwd+bwd" if train
            else "clips/s (10s@32x224x224) SlowFast-R50 feature extractor fwd",
            "value": round(value, 2), "unit": "clips/s", "n_gpus": world,
            "rccl_ranks": dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else None,
            **({"dist_backend": backend + " (test only: ranks share one GPU)"} if (share_gpu or backend != "nccl") else {}),
            "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": ("BASELINE configs[2]: SlowFast-R50 + 6-layer TxEnc verb-pred, "
                                    "fwd+bwd+Adam, dropout 0.1, 8 clips/GPU as 2 videos x 4 events" if train else
                                    "BASELINE configs[1]: SlowFast-R50 feature extractor only, eval, "
                                    "8 clips x 3x32x224x224 per GPU"),
                       "clips_per_gpu": CLIPS_PER_GPU, "hipgraph": used_graph,
                       **({"whatif": f"VS_WHATIF={os.environ['VS_WHATIF']}: kernel launches skipped, GARBAGE numerics -- a "
                                     "timing experiment, not a measurement of the workload"}
                          if os.environ.get("VS_WHATIF", "0") not in ("", "0") else {}),
                       **({} if CLIPS_PER_GPU == 8 else
                          {"probe": f"{CLIPS_PER_GPU} clips/GPU instead of the BASELINE config's 8: not the headline metric"}),
                       **({"probe_enc_layers": os.environ["VS_BENCH_ENC_LAYERS"]}
                          if train and os.environ.get("VS_BENCH_ENC_LAYERS", "6") != "6" else {}),
                       # the arithmetic of the line: bf16 operands (fp32 accumulation / statistics / optimizer); the
                       # logits' distance to the fp32 reference in that arithmetic, as last measured by the parity test
                       "parity": eval_parity_note(False if train else calibrated_main, train=train),
                       "grad_allreduce": (None if ts is None else
                                          (f"{len(ts.segments)} bucket(s), "
                                           f"{'bf16' if ts.grad_bf16 else 'fp32'} payload, "
                                           f"{'overlapped with backward' if ts.overlap else 'after backward'}"
                                           if ts.use_dist else "none (single process)")),
                       "model_tflops": round(value * flop_per_clip / 1e3, 2),
                       "frac_of_bf16_mfma_peak": round(value * flop_per_clip / 1e3 / world /
                                                       PEAK_BF16_TFLOPS, 4)},
            "roofline": roof, "cpu_baseline": cpu,
            **({"feat_fwd": fwd} if fwd is not None else {}),
            **({"canonical_b8x5": canon} if canon is not None else {}),
        }
        emit(line)
    if dist.is_available() and dist.is_initialized():
        dist.barrier()  # rank 0 is still in its instrumented pass / CPU baseline: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
