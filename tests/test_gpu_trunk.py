"""End-to-end parity of the HIP trunk / SFBase with the CPU fp32 oracle restatement
(oracle/slowfast_ref.py) on identical weights and inputs, at sizes the oracle finishes in
seconds.  bf16 activations between ~50 stacked layers: tolerances are on the relative L2
error of whole tensors (written next to each assert), stage by stage so a failure localises."""
import pytest
import torch

from gpu_utils import rel_err, rel_l2

pytestmark = pytest.mark.gpu


def _pair(arch, depth, width, frames, dev, seed=0):
    from oracle.slowfast_ref import VideoTrunk as RefTrunk, default_sf_cfg, randomize_bn
    from vidsitu_amd.trunk import VideoTrunk

    torch.manual_seed(seed)
    cfg = default_sf_cfg(arch, depth, width, frames)
    ref = RefTrunk(cfg)
    randomize_bn(ref, seed)
    ours = VideoTrunk(cfg)
    ours.load_state_dict(ref.state_dict(), strict=True)
    return cfg, ref, ours.to(dev)


def _ref_taps(ref, xs):
    taps = {}
    x = ref.s1(list(xs))
    if ref.multi:
        x = ref.s1_fuse(x)
    taps["s1"] = [t.clone() for t in x]
    for k in range(2, 6):
        x = getattr(ref, f"s{k}")(x)
        if ref.multi and k < 5:
            x = getattr(ref, f"s{k}_fuse")(x)
        if k == 2:
            x = [getattr(ref, f"pathway{p}_pool")(x[p]) for p in range(ref.num_pathways)]
        taps[f"s{k}"] = [t.clone() for t in x]
    return x, taps


def _inputs(cfg, n, hw, seed=1):
    g = torch.Generator().manual_seed(seed)
    t = cfg.DATA.NUM_FRAMES
    fast = torch.randn(n, 3, t, hw, hw, generator=g)
    if cfg.MODEL.ARCH == "slowfast":
        from oracle.slowfast_ref import slow_index

        return [fast.index_select(2, slow_index(t, cfg.SLOWFAST.ALPHA)), fast]
    return [fast]


@pytest.mark.parametrize("arch,depth,width,frames,n,hw", [
    ("i3d", "tiny", 8, 8, 2, 32),
    ("i3d", 50, 64, 8, 1, 64),
    ("slowfast", 50, 64, 32, 2, 64),
])
def test_trunk_eval_matches_oracle(arch, depth, width, frames, n, hw, dev):
    cfg, ref, ours = _pair(arch, depth, width, frames, dev)
    xs = _inputs(cfg, n, hw)
    ref.eval()
    ours.eval()
    with torch.no_grad():
        fr, taps_r = _ref_taps(ref, xs)
        ours.debug_taps = {}
        fo = ours.forward_features([x.to(dev) for x in xs])
    report = []
    for k in ["s1", "s2", "s3", "s4", "s5"]:
        for p, (a, b) in enumerate(zip(ours.debug_taps[k], taps_r[k])):
            assert tuple(a.shape) == tuple(b.shape), (k, p, a.shape, b.shape)
            report.append((k, p, rel_l2(a, b), rel_err(a, b)))
    print("\n".join(f"{k} pathway{p}: rel_l2 {l2:.3e} max {mx:.3e}" for k, p, l2, mx in report))
    for k, p, l2, mx in report:
        assert l2 < 3e-2, f"{k} pathway{p} rel_l2 {l2:.3e}"  # bf16 chain through <= 53 convs
    for a, b in zip(fo, fr):
        assert tuple(a.shape) == tuple(b.shape)  # logical NCDHW kept


def _soften_final_bn(model, seed):
    """gamma of each bottleneck's last BN in [0.1, 0.3] (the reference initialises it to 0,
    ZERO_INIT_FINAL_BN): keeps the random-weight net from amplifying rounding noise ~100x."""
    from torch import nn

    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, nn.BatchNorm3d) and getattr(m, "transform_final_bn", False):
            m.weight.data.copy_(0.1 + 0.2 * torch.rand(m.num_features, generator=g))


@pytest.mark.parametrize("arch,depth,width,frames,n,hw", [
    ("i3d", "tiny", 8, 8, 2, 32),
    ("slowfast", "mini", 64, 32, 2, 64),  # both pathways, 4 fusions, one block per stage
    ("slowfast", 50, 64, 32, 2, 64),
])
def test_trunk_train_step_matches_oracle(arch, depth, width, frames, n, hw, dev):
    """Train-mode forward (batch statistics, running-stat update) and the hand-written
    backward (parameter gradients) vs autograd on the oracle.

    Reference = the fp32 oracle with bf16 rounding applied where the kernels store bf16
    (oracle.slowfast_ref.emulate_bf16_storage): in train mode a random-weight ResNet
    amplifies 2^-9 rounding noise so strongly that the plain fp32 oracle moves by the same
    amount when only ITS storage is rounded (printed below for information)."""
    import copy

    from oracle.slowfast_ref import emulate_bf16_storage

    cfg, ref, ours = _pair(arch, depth, width, frames, dev, seed=3)
    _soften_final_bn(ref, 5)
    ours.load_state_dict(ref.state_dict(), strict=True)
    xs = _inputs(cfg, n, hw, seed=4)
    emu = copy.deepcopy(ref)
    emulate_bf16_storage(emu)
    ref.train()
    emu.train()
    ours.train()
    xs_b = [x.to(torch.bfloat16).float() for x in xs]
    fr = ref.forward_features(xs)
    fe = emu.forward_features(xs_b)
    g = torch.Generator().manual_seed(5)
    dfeat = [(torch.randn(f.shape, generator=g) / f.numel() ** 0.5).to(torch.bfloat16).float()
             for f in fr]
    sum((f * d).sum() for f, d in zip(fr, dfeat)).backward()
    sum((f * d).sum() for f, d in zip(fe, dfeat)).backward()
    fo = ours.forward_features([x.to(dev) for x in xs])
    for p, (a, b, c) in enumerate(zip(fo, fe, fr)):
        l2 = rel_l2(a, b)
        print(f"train fwd pathway{p}: vs bf16-emulating oracle {l2:.3e}; vs fp32 oracle "
              f"{rel_l2(a, c):.3e}; emulation vs fp32 {rel_l2(b, c):.3e}")
        # noise level: the emulation itself sits 1-3e-2 from the fp32 oracle here, and residual
        # differences (BN statistics from fp32 accumulators, summation order) are amplified alike
        assert l2 < 4e-2
    sum((f.float() * d.to(dev)).sum() for f, d in zip(fo, dfeat)).backward()
    # running statistics (fp32 partial sums of fp32 accumulators)
    sd_r, sd_o = emu.state_dict(), ours.state_dict()
    worst = 0.0
    for k in sd_r:
        if "running" in k:
            worst = max(worst, rel_err(sd_o[k], sd_r[k]))
        if "num_batches_tracked" in k:
            assert int(sd_o[k]) == int(sd_r[k]) == 1
    print(f"running stats worst max-normalised err {worst:.3e}")
    assert worst < 2e-2
    # parameter gradients
    pe, pr, po = dict(emu.named_parameters()), dict(ref.named_parameters()), dict(ours.named_parameters())
    rows, rows32 = [], []
    for k in pe:
        assert po[k].grad is not None, f"no gradient for {k}"
        rows.append((rel_l2(po[k].grad, pe[k].grad), k))
        rows32.append(rel_l2(pe[k].grad, pr[k].grad))
    rows.sort(reverse=True)
    print("worst parameter-gradient rel_l2 vs bf16-emulating oracle:\n" +
          "\n".join(f"  {e:.3e} {k}" for e, k in rows[:12]))
    med = sorted(e for e, _ in rows)[len(rows) // 2]
    print(f"median {med:.3e}   (for scale: emulation vs fp32 oracle median "
          f"{sorted(rows32)[len(rows32) // 2]:.3e}, max {max(rows32):.3e})")
    # Criterion: the HIP gradients sit inside the bf16-rounding noise band, i.e. closer to the
    # bf16-emulating oracle than that oracle is to plain fp32 (with batch-of-2 batch norm the
    # backward is chaotic: emulation vs fp32 is 0.13 / 0.18 / 0.33 median for the three nets).
    # The per-kernel tests (conv dgrad / wgrad, BN backward, pools) carry the tight bounds.
    med32 = sorted(rows32)[len(rows32) // 2]
    assert med < max(8e-2, 0.9 * med32), f"median parameter-gradient error {med:.3e} vs band {med32:.3e}"
    # the single worst parameter (the fast stem's BN weight, behind the whole backward) moves between 0.5 and 0.85
    # when nothing but the summation order of a batch-statistic partial changes (two builds of the small-channel
    # kernel, bitwise equal outputs): 2x the band's own maximum, and the 90th percentile inside the band's.
    # Tight evidence for the composed backward: test_gpu_parity_full.py (layer-local, no amplification).
    p90 = sorted(e for e, _ in rows)[int(len(rows) * 0.9)]
    assert p90 < max(2.5e-1, 1.2 * sorted(rows32)[int(len(rows32) * 0.9)]), f"90th percentile {p90:.3e}"
    assert rows[0][0] < max(3.5e-1, 2.0 * max(rows32)), f"worst parameter gradient {rows[0]}"


def test_sfbase_logits_and_top5_indices(dev):
    """SFBase end to end (mdl_sf_base.py:213-216) + EvalB top-5 (evl_vsitu.py:39-42): logits
    within tolerance of the oracle and verb indices bit-exact wherever the oracle's own
    top-5 margins exceed that tolerance (otherwise index equality is ill-posed)."""
    from oracle.slowfast_ref import SFBaseRef, randomize_bn
    from vidsitu_amd import synth_data
    from vidsitu_amd.evl_vsitu import EvalB
    from vidsitu_amd.extended_config import get_cfg
    from vidsitu_amd.mdl_selector import get_mdl_loss_eval

    cfg = get_cfg({"mdl.sf_mdl_name": "i3d_tiny", "synth.num_verbs": 97})
    comm = synth_data.make_comm(cfg)
    torch.manual_seed(0)
    sel = get_mdl_loss_eval(cfg)
    mdl = sel["mdl"](cfg=cfg, comm=comm)
    ref = SFBaseRef(cfg.sf_mdl, 97)
    randomize_bn(ref, 1)
    with torch.no_grad():
        for lin in (ref.proj_head[0], ref.proj_head[2]):
            lin.weight.normal_(0, 0.2)
    mdl.load_state_dict(ref.state_dict(), strict=True)
    mdl = mdl.to(dev).eval()
    ref.eval()
    batch = synth_data.synth_batch(cfg, comm, bs=2, n_ev=5, crop=32)
    with torch.no_grad():
        lr = ref([batch["frms_ev_fast_tensor"].flatten(0, 1)]).view(2, 5, -1)
        gbatch = {k: v.to(dev) for k, v in batch.items()}
        lo = mdl(gbatch)["mdl_out"]
    assert tuple(lo.shape) == (2, 5, 97)
    err = float((lo.cpu() - lr).abs().max())
    print(f"logits max abs err {err:.3e} (max |logit| {float(lr.abs().max()):.3f})")
    assert err < 2e-2 * float(lr.abs().max())
    out = EvalB(cfg, comm, dev).forward_one_batch(mdl, gbatch)
    srt, ix = lr.sort(dim=-1, descending=True)
    for b in range(2):
        for e in range(5):
            margins = (srt[b, e, :5] - srt[b, e, 1:6])
            if float(margins.min()) > 2.5 * err:
                assert out[b]["pred_ixs_ev"][e] == ix[b, e, :5].tolist()
    # the loss / backward path of the plugin surface runs end to end
    mdl.train()
    loss = sel["loss"](cfg, comm)(mdl(gbatch), gbatch)["loss"]
    loss.backward()
    assert torch.isfinite(loss) and mdl.proj_head[0].weight.grad is not None
    assert mdl.sf_mdl.s1.pathway0_stem.conv.weight.grad is not None


def test_feature_dump_roundtrip_into_txenc(dev, tmp_path):
    """A8 (`feat_extractor.py:90-112`): trunk -> head -> `<vseg>_feats.npy` f32 [5, D] per video,
    equal to the oracle's features, readable by the A9-A11 consumer (`dat_loader.py:503-511` ->
    `SFPreFeats_TxEncDec.forward_encoder`, `mdl_sf_base.py:806-832`)."""
    import numpy as np
    from oracle.slowfast_ref import SFBaseRef, randomize_bn
    from vidsitu_amd import feat_extractor as fx, synth_data
    from vidsitu_amd.extended_config import get_cfg
    from vidsitu_amd.mdl_selector import get_mdl_loss_eval

    cfg = get_cfg({"mdl.sf_mdl_name": "i3d_tiny", "synth.num_verbs": 31,
                   "ds.vsitu.vsitu_frm_feats": str(tmp_path)})
    comm = synth_data.make_comm(cfg)
    torch.manual_seed(0)
    mdl = get_mdl_loss_eval(cfg)["mdl"](cfg=cfg, comm=comm)
    ref = SFBaseRef(cfg.sf_mdl, 31)
    randomize_bn(ref, 3)
    mdl.load_state_dict(ref.state_dict(), strict=True)
    mdl = mdl.to(dev).eval()
    ref.eval()
    ds = fx.SynthFrameDataset(cfg, comm, n_videos=3, n_ev=5, seed=11, crop=32)
    ext = fx.FeatExtract(cfg)
    ext.set_mdl_dl(mdl, fx.SimpleLoader(ds, batch_size=2), mdl_name="i3d_tiny_synth", split_name="valid")
    files = ext.forward_all(device=dev)
    assert [f.name for f in files] == [f"{n}_feats.npy" for n in ds.vseg_lst]
    for ix, f in enumerate(files):
        arr = np.load(f)
        assert arr.dtype == np.float32 and arr.shape[0] == 5 and arr.ndim == 2
        with torch.no_grad():
            want = ref.forward_feats([ds[ix]["frms_ev_fast_tensor"]])
        want = want.reshape(5, -1)
        assert arr.shape == tuple(want.shape)
        err = float(np.abs(arr - want.numpy()).max()) / float(want.abs().max())
        assert err < 3e-2, f"video {ix}: feature error {err:.3e}"
    # consumer side: [B, 5, D] features -> vid_feat_encoder -> TxEncoder -> EncoderOut [1, 5B, 1024]
    from vidsitu_amd.mdl_sf_base import SFPreFeats_TxEncDec
    enc = SFPreFeats_TxEncDec(cfg, comm, head_dim=np.load(files[0]).shape[1]).to(dev).eval()
    feats = torch.stack([fx.read_frm_feats(ext.out_tdir, n)["frm_feats"] for n in ds.vseg_lst]).to(dev)
    with torch.no_grad():
        out = enc.forward_encoder({"frm_feats": feats, "vseg_idx": torch.arange(3, device=dev)})
    assert tuple(out.encoder_out.shape) == (1, 15, 1024) and torch.isfinite(out.encoder_out).all()


def test_uint8_frames_path_is_bitwise_the_fp32_contract(dev):
    """f1 (second half): uint8 RGB frames normalised / packed / slow-gathered on the GPU give the
    same bf16 stem inputs, hence the same logits, bit for bit, as the reference's fp32 tensors."""
    from vidsitu_amd import ops, synth_data
    from vidsitu_amd.extended_config import get_cfg
    from vidsitu_amd.mdl_selector import get_mdl_loss_eval

    cfg = get_cfg({"mdl.sf_mdl_name": "slow_fast_mini", "synth.num_verbs": 23})
    comm = synth_data.make_comm(cfg)
    torch.manual_seed(0)
    mdl = get_mdl_loss_eval(cfg)["mdl"](cfg=cfg, comm=comm).to(dev).eval()
    b8 = synth_data.synth_u8_batch(cfg, comm, bs=1, n_ev=2, crop=32)
    ref = synth_data.reference_tensors(b8, cfg, comm)
    fr = b8["frms_ev_fast_u8"].flatten(0, 1).to(dev)
    # kernel level: both pathways, both packed layouts
    idx = synth_data.slow_index(fr.shape[1], cfg.sf_mdl.SLOWFAST.ALPHA).to(torch.int32).to(dev)
    for cpad in (4, 8):
        got_f = ops.frames_u8_pack(fr, cpad)
        got_s = ops.frames_u8_pack(fr, cpad, idx)
        want_f = ops.pack_input(ref["frms_ev_fast_tensor"].flatten(0, 1).to(dev), cpad)
        want_s = ops.pack_input(ref["frms_ev_slow_tensor"].flatten(0, 1).to(dev), cpad)
        assert torch.equal(got_f.view(torch.int16), want_f.view(torch.int16))
        assert torch.equal(got_s.view(torch.int16), want_s.view(torch.int16))
    # model level
    with torch.no_grad():
        a = mdl({k: v.to(dev) for k, v in b8.items()})["mdl_out"]
        b = mdl({k: v.to(dev) for k, v in ref.items()})["mdl_out"]
    assert torch.equal(a, b)


def test_every_gradient_is_overwritten(dev):
    """bench.py skips the per-step gradient memset: poison every parameter's gradient with NaN and
    check that one backward pass of the config-3 model (SlowFast mini + TxEncoder) leaves none."""
    from vidsitu_amd import synth_data
    from vidsitu_amd.extended_config import get_cfg
    from vidsitu_amd.mdl_selector import get_mdl_loss_eval
    from vidsitu_amd.optim import ParamArena

    cfg = get_cfg({"mdl.mdl_name": "sf_base_txenc", "mdl.sf_mdl_name": "slow_fast_mini",
                   "synth.num_verbs": 31, "tx_dec.encoder_layers": 2})
    comm = synth_data.make_comm(cfg)
    torch.manual_seed(0)
    sel = get_mdl_loss_eval(cfg)
    mdl = sel["mdl"](cfg=cfg, comm=comm).to(dev).train()
    arena = ParamArena(mdl)
    batch = synth_data.synth_batch(cfg, comm, bs=2, n_ev=2, crop=32, device=dev, dtype=torch.bfloat16)
    loss_fn = sel["loss"](cfg, comm)
    for it in range(2):
        arena.zero_grad(fill=(it == 0))
        for p in arena.params:
            p.grad.fill_(float("nan"))
        loss_fn(mdl(batch), batch)["loss"].backward()
        torch.cuda.synchronize()
        bad = [n for n, p in mdl.named_parameters() if not torch.isfinite(p.grad).all()]
        assert not bad, f"gradients not overwritten: {bad[:5]}"
        assert torch.isfinite(arena.grad).all()


@pytest.mark.parametrize("sf_name,crop,bs,n_ev,layers", [("slow_fast_mini", 64, 2, 2, 2),
                                                         ("slow_fast_nl_r50_8x8", 224, 2, 4, 6)],
                         ids=["mini", "r50_bench_config"])
def test_hipgraph_two_stream_step_is_bitwise_the_one_stream_eager_step(sf_name, crop, bs, n_ev, layers, dev):
    """The timed region of bench.py replays a hipGraph whose pathway / wgrad branches run on
    parallel streams.  Every kernel is deterministic, so gradients and loss of that replay must
    equal, bit for bit, an eager step with everything on one stream -- a missing dependency between
    the streams would show up here as a mismatch."""
    from vidsitu_amd import synth_data, trunk as T
    from vidsitu_amd.extended_config import get_cfg
    from vidsitu_amd.mdl_selector import get_mdl_loss_eval
    from vidsitu_amd.optim import ParamArena

    cfg = get_cfg({"mdl.mdl_name": "sf_base_txenc", "mdl.sf_mdl_name": sf_name,
                   "synth.num_verbs": 31, "tx_dec.encoder_layers": layers, "tx_dec.dropout": 0.0})
    comm = synth_data.make_comm(cfg)
    torch.manual_seed(0)
    sel = get_mdl_loss_eval(cfg)
    mdl = sel["mdl"](cfg=cfg, comm=comm).to(dev).train()
    loss_fn = sel["loss"](cfg, comm)
    arena = ParamArena(mdl)
    batch = synth_data.synth_batch(cfg, comm, bs=bs, n_ev=n_ev, crop=crop, device=dev, dtype=torch.bfloat16)
    out = {}

    def step():
        arena.zero_grad()
        loss = loss_fn(mdl(batch), batch)["loss"]
        loss.backward()
        out["loss"] = loss.detach()

    saved = (T.VideoTrunk.dual_stream, T._WgradLanes.enabled)
    try:
        T.VideoTrunk.dual_stream, T._WgradLanes.enabled = False, False
        step()
        torch.cuda.synchronize()
        g_ref, l_ref = arena.grad.clone(), out["loss"].clone()
        assert torch.isfinite(g_ref).all() and float(g_ref.abs().max()) > 0
        T.VideoTrunk.dual_stream, T._WgradLanes.enabled = True, True
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        assert torch.equal(arena.grad, g_ref), "eager two-stream step differs from one-stream"
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step()
        for it in range(3):
            arena.grad.fill_(float("nan"))
            graph.replay()
            torch.cuda.synchronize()
            if not torch.equal(arena.grad, g_ref):
                bad = []
                for (n, p_), off, end in zip(mdl.named_parameters(), arena.offsets, arena.offsets[1:]):
                    a, b = arena.grad[off:off + p_.numel()], g_ref[off:off + p_.numel()]
                    if not torch.equal(a, b):
                        bad.append((n, float((a - b).abs().max()), float(b.abs().max()),
                                    int(torch.isnan(a).sum())))
                raise AssertionError(f"hipGraph replay {it} differs from the eager step in "
                                     f"{len(bad)} parameters: {bad[:12]}")
            assert torch.equal(out["loss"], l_ref)
        # the distributed step of bench.py: trunk backward deferred out of autograd and captured as
        # one graph per segment (gradient buckets are all-reduced between the replays)
        trunk = mdl.sf_mdl
        trunk.defer_backward = True
        segs = [step] + [(lambda sg=sg: trunk.run_backward_segment(sg)) for sg in trunk.BWD_SEGMENTS]
        for fn in segs:  # eager warm-up of the deferred path
            fn()
        torch.cuda.synchronize()
        assert torch.equal(arena.grad, g_ref), "eager segmented step differs"
        graphs, pool = [], None
        for fn in segs:
            gseg = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gseg, pool=pool):
                fn()
            pool = gseg.pool()
            graphs.append(gseg)
        for it in range(2):
            arena.grad.fill_(float("nan"))
            for gseg in graphs:
                gseg.replay()
            torch.cuda.synchronize()
            assert torch.equal(arena.grad, g_ref), f"segmented hipGraph replay {it} differs"
    finally:
        T.VideoTrunk.dual_stream, T._WgradLanes.enabled = saved
        mdl.sf_mdl.defer_backward = False


def test_hipgraph_two_stream_eval_forward_is_bitwise_the_one_stream_one(dev):
    """configs[1] (feature extractor, eval): activations are freed as the pass goes, so a buffer
    re-used across the two pathway streams would corrupt the features -- compare the two-stream
    hipGraph replay with a one-stream eager pass, bit for bit, at the bench configuration."""
    from vidsitu_amd import synth_data, trunk as T
    from vidsitu_amd.extended_config import get_cfg
    from vidsitu_amd.mdl_selector import get_mdl_loss_eval

    cfg = get_cfg({"mdl.mdl_name": "sf_base"})
    comm = synth_data.make_comm(cfg)
    torch.manual_seed(0)
    mdl = get_mdl_loss_eval(cfg)["mdl"](cfg=cfg, comm=comm).to(dev).eval()
    batch = synth_data.synth_batch(cfg, comm, bs=2, n_ev=4, device=dev, dtype=torch.bfloat16)
    out = {}

    def fwd():
        with torch.no_grad():
            out["f"] = mdl.head(mdl.forward_encoder(batch))

    saved = T.VideoTrunk.dual_stream
    try:
        T.VideoTrunk.dual_stream = False
        fwd()
        torch.cuda.synchronize()
        ref = out["f"].clone()
        T.VideoTrunk.dual_stream = True
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                fwd()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        assert torch.equal(out["f"], ref)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fwd()
        for _ in range(3):
            out["f"].fill_(float("nan"))
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(out["f"], ref)
    finally:
        T.VideoTrunk.dual_stream = saved


def test_bench_style_training_trajectory_in_hipgraph_matches_eager(dev):
    """Three optimizer steps replayed from the bench's graph (two pathway streams, wgrad lane,
    dgrad weight images refreshed on a third stream at step start, bf16 cast fused into Adam)
    leave exactly the parameters of three plain eager steps with a full weight refresh."""
    from vidsitu_amd import synth_data, trunk as T
    from vidsitu_amd.extended_config import get_cfg
    from vidsitu_amd.mdl_selector import get_mdl_loss_eval
    from vidsitu_amd.optim import ArenaAdam, ParamArena

    cfg = get_cfg({"mdl.mdl_name": "sf_base_txenc", "mdl.sf_mdl_name": "slow_fast_mini",
                   "synth.num_verbs": 31, "tx_dec.encoder_layers": 2, "tx_dec.dropout": 0.0})
    comm = synth_data.make_comm(cfg)
    torch.manual_seed(0)
    sel = get_mdl_loss_eval(cfg)
    mdl = sel["mdl"](cfg=cfg, comm=comm).to(dev).train()
    loss_fn = sel["loss"](cfg, comm)
    arena = ParamArena(mdl)
    opt = ArenaAdam(arena, lr=1e-2)
    batch = synth_data.synth_batch(cfg, comm, bs=2, n_ev=2, crop=64, device=dev, dtype=torch.bfloat16)
    init = arena.data.clone()
    bufs = {k: v.clone() for k, v in mdl.named_buffers()}

    def reset():
        arena.data.copy_(init)
        opt.m.zero_(); opt.v.zero_(); opt.t.zero_()
        for k, v in mdl.named_buffers():
            v.copy_(bufs[k])
        arena.refresh()

    def fwd_bwd():
        opt.zero_grad()
        loss_fn(mdl(batch), batch)["loss"].backward()

    saved = (T.VideoTrunk.dual_stream, T._WgradLanes.enabled)
    try:
        T.VideoTrunk.dual_stream, T._WgradLanes.enabled = False, False
        reset()
        for _ in range(3):
            fwd_bwd()
            opt.step()
        torch.cuda.synchronize()
        want = arena.data.clone()
        assert not torch.equal(want, init)
        T.VideoTrunk.dual_stream, T._WgradLanes.enabled = True, True

        def step():
            arena.transposes_async()
            fwd_bwd()
            arena._join_transposes()
            opt.step(defer_transposes=True)

        reset()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step()  # warm-up (allocations, lane streams)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        torch.cuda.synchronize()
        reset()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        assert torch.equal(arena.data, want), \
            f"max |diff| {float((arena.data - want).abs().max()):.3e}"
    finally:
        T.VideoTrunk.dual_stream, T._WgradLanes.enabled = saved


def test_overfit_one_batch_end_to_end(dev):
    """The reference's `overfit_batch` mode (`trn_utils.py:915-939`): the whole plugin surface --
    SlowFast trunk (both pathways) + TxEncoder + loss + manual backward + fused Adam -- drives the
    loss of one fixed batch towards zero and its top-1 accuracy to 1."""
    from vidsitu_amd import synth_data
    from vidsitu_amd.extended_config import get_cfg
    from vidsitu_amd.mdl_selector import get_mdl_loss_eval
    from vidsitu_amd.optim import ArenaAdam, ParamArena

    cfg = get_cfg({"mdl.mdl_name": "sf_base_txenc", "synth.num_verbs": 31, "tx_dec.encoder_layers": 2})
    comm = synth_data.make_comm(cfg)
    torch.manual_seed(0)
    sel = get_mdl_loss_eval(cfg)
    mdl = sel["mdl"](cfg=cfg, comm=comm).to(dev).train()  # SlowFast-R50
    loss_fn = sel["loss"](cfg, comm)
    arena = ParamArena(mdl)
    opt = ArenaAdam(arena, lr=3e-4)
    batch = synth_data.synth_batch(cfg, comm, bs=2, n_ev=4, crop=112, device=dev, dtype=torch.bfloat16)
    losses = []
    for _ in range(40):
        opt.zero_grad()
        loss = loss_fn(mdl(batch), batch)["loss"]
        loss.backward()
        opt.step()
        losses.append(loss.detach())
    losses = [float(x) for x in losses]
    assert losses[-1] < 0.05 * losses[0], (losses[0], losses[-1])
    mdl.eval()
    _, mets = sel["evl"](cfg, comm, dev)(mdl, loss_fn, [batch])
    assert mets["Per_Ev_Top_1"] >= 0.9, mets


@pytest.mark.parametrize("arch,bwd", [("slowfast", False), ("slowfast", True), ("i3d", True)])
def test_fused_stem_pool_step_is_bitwise_the_unfused_step(arch, bwd, dev):
    """`ResNetBasicStem.fuse_pool`: features and every gradient of a train-mode pass are bit for bit those of the pass
    with bn_apply, maxpool_hw and maxpool_hw_bwd as separate launches."""
    from oracle.slowfast_ref import default_sf_cfg
    from vidsitu_amd import trunk as T

    torch.manual_seed(3)
    frames = 32 if arch == "slowfast" else 8
    cfg = default_sf_cfg(arch, 50, 64, frames)
    mdl = T.VideoTrunk(cfg).to(dev).train()
    g = torch.Generator().manual_seed(4)
    fast = torch.randn(2, 3, frames, 64, 64, generator=g).to(dev)
    xs = [fast[:, :, ::4].contiguous(), fast] if arch == "slowfast" else [fast]
    bufs = {k: v.clone() for k, v in mdl.named_buffers()}

    calls = []
    orig = T.ops.bn_apply_maxpool

    def run(fuse):
        T.ResNetBasicStem.fuse_pool, T.ResNetBasicStem.fuse_pool_bwd = fuse, bwd
        T.ops.bn_apply_maxpool = lambda *a, **k: (calls.append(fuse), orig(*a, **k))[1]
        for k, v in mdl.named_buffers():
            v.copy_(bufs[k])
        for p in mdl.parameters():
            p.grad = None
        feats = mdl.forward_features([x.clone() for x in xs])
        gg = torch.Generator().manual_seed(5)
        sum((f.float() * torch.randn(f.shape, generator=gg).to(dev)).sum() for f in feats).backward()
        torch.cuda.synchronize()
        return [f.detach().clone() for f in feats], {k: p.grad.clone() for k, p in mdl.named_parameters()}

    try:
        f0, g0 = run(False)
        f1, g1 = run(True)
    finally:
        T.ResNetBasicStem.fuse_pool, T.ResNetBasicStem.fuse_pool_bwd = True, False
        T.ops.bn_apply_maxpool = orig
    assert calls == [True] * len(xs)  # one fused pass per stem, only in the fused run
    assert all(torch.equal(a, b) for a, b in zip(f0, f1))
    bad = [k for k in g0 if not torch.equal(g0[k], g1[k])]
    assert not bad, bad[:5]


def test_pair_launch_of_dgrad_and_wgrad_is_bitwise_the_separate_launches(dev):
    """`_Unit.pair_launch`: a unit's data gradient and weight gradient as one launch (csrc/conv_pair.hip) where both run
    on the 128 x 128 ring kernels -- features and every gradient of a SlowFast-R50 train-mode pass at 112^2 bit for
    bit those of the pass with separate launches; and pairs are actually issued."""
    from oracle.slowfast_ref import default_sf_cfg
    from vidsitu_amd import ops, trunk as T

    torch.manual_seed(3)
    cfg = default_sf_cfg("slowfast", 50, 64, 32)
    mdl = T.VideoTrunk(cfg).to(dev).train()
    g = torch.Generator().manual_seed(4)
    fast = torch.randn(2, 3, 32, 112, 112, generator=g).to(dev)
    xs = [fast[:, :, ::4].contiguous(), fast]
    bufs = {k: v.clone() for k, v in mdl.named_buffers()}

    def run(pair):
        T._Unit.pair_launch = pair
        T.ResBlock.group_wgrads = False  # (round 5: grouped blocks launch their weight gradients together, not in pairs)
        for k, v in mdl.named_buffers():
            v.copy_(bufs[k])
        for p in mdl.parameters():
            p.grad = None
        n0 = ops.conv_pair_count()
        feats = mdl.forward_features([x.clone() for x in xs])
        gg = torch.Generator().manual_seed(5)
        sum((f.float() * torch.randn(f.shape, generator=gg).to(dev)).sum() for f in feats).backward()
        torch.cuda.synchronize()
        return ([f.detach().clone() for f in feats], {k: p.grad.clone() for k, p in mdl.named_parameters()},
                ops.conv_pair_count() - n0)

    try:
        f0, g0, n_off = run(False)
        f1, g1, n_on = run(True)
    finally:
        T._Unit.pair_launch = True
        T.ResBlock.group_wgrads = True
    assert n_off == 0 and n_on >= 1, (n_off, n_on)
    assert all(torch.equal(a, b) for a, b in zip(f0, f1))
    bad = [k for k in g0 if not torch.equal(g0[k], g1[k])]
    assert not bad, bad[:5]


def test_deferred_slab_reduce_merged_with_the_next_finalize_is_bitwise_and_shorter(dev, monkeypatch):
    """Round 3: the slab reduce behind a weight gradient waits for the next unit's BN-backward finalize and shares its
    launch (vs_wgrad_reduce_defer, bn_bwd_finalize_wgrad_reduce_kernel).  Same bodies: every gradient of a SlowFast-R50
    step bit for bit, with fewer launches; nothing is left pending after a backward pass."""
    from vidsitu_amd import ops, synth_data
    from vidsitu_amd.extended_config import get_cfg
    from vidsitu_amd.mdl_selector import get_mdl_loss_eval
    from vidsitu_amd.optim import ArenaAdam, ParamArena
    from vidsitu_amd.train_step import TrainStep

    cfg = get_cfg({"mdl.mdl_name": "sf_base_txenc", "mdl.sf_mdl_name": "slow_fast_nl_r50_8x8", "synth.num_verbs": 31,
                   "tx_dec.encoder_layers": 2, "tx_dec.dropout": 0.0})
    comm = synth_data.make_comm(cfg)
    torch.manual_seed(0)
    sel = get_mdl_loss_eval(cfg)
    mdl = sel["mdl"](cfg=cfg, comm=comm).to(dev).train()
    arena = ParamArena(mdl)
    opt = ArenaAdam(arena, lr=1e-3)
    batch = synth_data.synth_batch(cfg, comm, bs=2, n_ev=2, crop=96, device=dev, dtype=torch.bfloat16)
    bufs = {k: v.clone() for k, v in mdl.named_buffers()}
    lib = ops._lib.load()
    out = {}
    for merge in (False, True):
        monkeypatch.setattr(ops, "REDUCE_MERGE", merge)
        for deferred in (False, True):  # the autograd-driven backward and the segment-by-segment one
            for k, v in mdl.named_buffers():
                v.copy_(bufs[k])
            ts = TrainStep(mdl, sel["loss"](cfg, comm), arena, opt, batch, world=1, use_dist=False, overlap=deferred)
            ts.overlap = deferred
            ts.segments = ts._build_segments()
            arena.grad.fill_(float("nan"))
            n0 = lib.vs_launch_count()
            for fn, _ in ts.segments:
                fn()
            torch.cuda.synchronize()
            out[(merge, deferred)] = (arena.grad.clone(), lib.vs_launch_count() - n0)
    mdl.sf_mdl.defer_backward = False
    for deferred in (False, True):
        g0, n_sep = out[(False, deferred)]
        g1, n_mrg = out[(True, deferred)]
        assert torch.isfinite(g0).all()
        assert torch.equal(g0, g1), f"deferred={deferred}: gradients differ with the merged launches"
        assert n_mrg <= n_sep - 30, (n_sep, n_mrg)
    assert torch.equal(out[(True, False)][0], out[(True, True)][0])


def test_eval_fused_bc_blocks_match_the_two_launch_blocks_and_save_thirteen_launches(dev):
    """configs[1] (feature extractor, eval): conv b -> conv c of the fast pathway's res2 / res3 / res4 bottlenecks as one launch
    (ResBlock.fuse_bc, vs_conv_fwd_bc) against the same forward with two launches per block -- 3 + 4 + 6 launches fewer
    on SlowFast-R50, pooled features equal to a bf16 ulp of the block outputs."""
    from vidsitu_amd import _lib, synth_data, trunk as T
    from vidsitu_amd.extended_config import get_cfg
    from vidsitu_amd.mdl_selector import get_mdl_loss_eval

    cfg = get_cfg({"mdl.mdl_name": "sf_base"})
    comm = synth_data.make_comm(cfg)
    torch.manual_seed(0)
    mdl = get_mdl_loss_eval(cfg)["mdl"](cfg=cfg, comm=comm).to(dev).eval()
    batch = synth_data.synth_batch(cfg, comm, bs=1, n_ev=2, device=dev, dtype=torch.bfloat16)
    lib = _lib.load()
    got = {}
    saved = T.ResBlock.fuse_bc
    try:
        for fuse in (False, True):
            T.ResBlock.fuse_bc = fuse
            with torch.no_grad():
                mdl.head(mdl.forward_encoder(batch))  # warm-up: folds, weight casts
                torch.cuda.synchronize()
                n0 = lib.vs_launch_count()
                f = mdl.head(mdl.forward_encoder(batch))
            torch.cuda.synchronize()
            got[fuse] = (f.float().clone(), lib.vs_launch_count() - n0)
    finally:
        T.ResBlock.fuse_bc = saved
    assert got[False][1] - got[True][1] == 13, (got[False][1], got[True][1])
    err = (got[True][0] - got[False][0]).abs().max().item() / got[False][0].abs().max().item()
    assert err < 2e-3, err


def test_apply_on_load_train_step_is_bitwise_and_launches_fewer_kernels(dev):
    """`ResBlock.aol` (VS_TRAIN_AOL=1): the b unit of a slow-pathway bottleneck stops behind its finalize and the c unit's
    forward and weight gradient form relu(y * scale + shift) on their operand fragments (vs_conv_fwd_aol /
    vs_conv_wgrad_aol) -- features, running statistics and every gradient of a SlowFast-R50 train-mode pass at 224^2
    bit for bit those of the pass that stores the activation, with one vs_bn_apply launch fewer per such block."""
    from oracle.slowfast_ref import default_sf_cfg
    from vidsitu_amd import _lib, trunk as T

    torch.manual_seed(3)
    cfg = default_sf_cfg("slowfast", 50, 64, 32)
    mdl = T.VideoTrunk(cfg).to(dev).train()
    g = torch.Generator().manual_seed(4)
    fast = torch.randn(2, 3, 32, 224, 224, generator=g).to(dev)
    xs = [fast[:, :, ::4].contiguous(), fast]
    bufs = {k: v.clone() for k, v in mdl.named_buffers()}
    lib = _lib.load()

    def run(aol):
        T.ResBlock.aol = aol
        for k, v in mdl.named_buffers():
            v.copy_(bufs[k])
        for p in mdl.parameters():
            p.grad = None
        torch.cuda.synchronize()
        n0 = lib.vs_launch_count()
        feats = mdl.forward_features([x.clone() for x in xs])
        gg = torch.Generator().manual_seed(5)
        sum((f.float() * torch.randn(f.shape, generator=gg).to(dev)).sum() for f in feats).backward()
        torch.cuda.synchronize()
        return ([f.detach().clone() for f in feats], {k: p.grad.clone() for k, p in mdl.named_parameters()},
                {k: v.clone() for k, v in mdl.named_buffers()}, lib.vs_launch_count() - n0)

    saved = T.ResBlock.aol
    T.ResBlock.group_wgrads = False  # (apply-on-load blocks keep per-unit weight gradients: compare like with like)
    try:
        f0, g0, b0, n_off = run(False)
        f1, g1, b1, n_on = run(True)
    finally:
        T.ResBlock.aol = saved
        T.ResBlock.group_wgrads = True
    assert n_off - n_on >= 4, (n_off, n_on)  # at 2 clips the s2 / s3 blocks' plans have the transform
    assert all(torch.equal(a, b) for a, b in zip(f0, f1))
    bad = [k for k in g0 if not torch.equal(g0[k], g1[k])]
    assert not bad, bad[:5]
    bad = [k for k in b0 if not torch.equal(b0[k], b1[k])]
    assert not bad, bad[:5]


def test_shortcut_apply_inside_the_last_apply_pass_is_bitwise_and_saves_eight_launches(dev):
    """`ResBlock.fuse_sc_apply` (round 5): the shortcut unit's BN is applied inside the c unit's apply pass
    (vs_bn_apply2) -- features, running statistics and every gradient of a SlowFast-R50 train-mode pass bit for bit
    those of the pass with the shortcut's own apply launch, eight launches (the res0 blocks of four stages x two
    pathways) fewer."""
    from oracle.slowfast_ref import default_sf_cfg
    from vidsitu_amd import ops, trunk as T

    torch.manual_seed(3)
    cfg = default_sf_cfg("slowfast", 50, 64, 32)
    mdl = T.VideoTrunk(cfg).to(dev).train()
    g = torch.Generator().manual_seed(4)
    fast = torch.randn(2, 3, 32, 96, 96, generator=g).to(dev)
    xs = [fast[:, :, ::4].contiguous(), fast]
    bufs = {k: v.clone() for k, v in mdl.named_buffers()}
    lib = ops._lib.load()

    def run(fuse, fuse_bwd=False):
        T.ResBlock.fuse_sc_apply, T.ResBlock.fuse_sc_bwd = fuse, fuse_bwd
        for k, v in mdl.named_buffers():
            v.copy_(bufs[k])
        for p in mdl.parameters():
            p.grad = None
        n0 = lib.vs_launch_count()
        feats = mdl.forward_features([x.clone() for x in xs])
        n_fwd = lib.vs_launch_count() - n0
        gg = torch.Generator().manual_seed(5)
        sum((f.float() * torch.randn(f.shape, generator=gg).to(dev)).sum() for f in feats).backward()
        torch.cuda.synchronize()
        return ([f.detach().clone() for f in feats], {k: p.grad.clone() for k, p in mdl.named_parameters()},
                {k: v.clone() for k, v in mdl.named_buffers()}, n_fwd)

    try:
        run(True)  # (the first pass also builds the bf16 weight images)
        f0, g0, b0, n0 = run(False)
        f1, g1, b1, n1 = run(True)
        # ... and the backward counterpart (`ResBlock.fuse_sc_bwd`, vs_bn_bwd_apply2): the c unit's and the shortcut
        # unit's BN-backward apply as one pass over the block's output gradient
        nb0 = lib.vs_launch_count()
        f2, g2, b2, _ = run(True, True)
        nb2 = lib.vs_launch_count() - nb0
        nb0 = lib.vs_launch_count()
        run(True, False)
        nb1 = lib.vs_launch_count() - nb0
    finally:
        T.ResBlock.fuse_sc_apply, T.ResBlock.fuse_sc_bwd = True, True
    assert n0 - n1 == 8, (n0, n1)
    # (one apply launch fewer per res0 block; the shortcut's finalize no longer carries conv a's pending slab reduce,
    #  which then takes a launch of its own: never more launches, 0.29 GB per step fewer)
    assert nb2 <= nb1, (nb1, nb2)
    assert all(torch.equal(a, b) for a, b in zip(f0, f1)) and all(torch.equal(a, b) for a, b in zip(f0, f2))
    assert not [k for k in g0 if not torch.equal(g0[k], g1[k])]
    assert not [k for k in g0 if not torch.equal(g0[k], g2[k])]
    assert not [k for k in b0 if not torch.equal(b0[k], b1[k])]


@pytest.mark.parametrize("sf_name,crop", [("slow_fast_mini", 64), ("i3d_tiny", 64), ("i3d_r50_nl_8x8", 64)])
def test_weight_rounding_calibration_on_every_trunk_family(sf_name, crop, dev):
    """`SFBase.calibrate_weight_rounding` on the two-pathway, single-pathway and non-local trunks: the eval logits move
    TOWARDS the fp32 oracle's (calibration on other clips), the correction survives repeated forwards bit for bit, is
    dropped by `reset_weight_rounding` and by a weight update, and never touches the training path."""
    from oracle.slowfast_ref import SFBaseRef, randomize_bn
    from vidsitu_amd import synth_data
    from vidsitu_amd.extended_config import get_cfg
    from vidsitu_amd.mdl_selector import get_mdl_loss_eval

    cfg = get_cfg({"mdl.sf_mdl_name": sf_name, "synth.num_verbs": 64})
    comm = synth_data.make_comm(cfg)
    torch.manual_seed(0)
    mdl = get_mdl_loss_eval(cfg)["mdl"](cfg=cfg, comm=comm)
    ref = SFBaseRef(cfg.sf_mdl, 64)
    randomize_bn(ref, 1)
    mdl.load_state_dict(ref.state_dict(), strict=True)
    mdl, ref = mdl.to(dev).eval(), ref.eval()
    batch = synth_data.synth_batch(cfg, comm, bs=1, n_ev=2, crop=crop, seed=11)
    cal = synth_data.synth_batch(cfg, comm, bs=1, n_ev=2, crop=crop, seed=12)
    gb = {k: v.to(dev) for k, v in batch.items()}
    with torch.no_grad():
        inp = [batch["frms_ev_fast_tensor"].flatten(0, 1)]
        if "frms_ev_slow_tensor" in batch:
            inp = [batch["frms_ev_slow_tensor"].flatten(0, 1)] + inp
        want = ref(inp).view(1, 2, -1)
        plain = mdl(gb)["mdl_out"].float().cpu()
        n = mdl.calibrate_weight_rounding({k: v.to(dev) for k, v in cal.items()})
        assert n >= 8
        got = mdl(gb)["mdl_out"].float().cpu()
        again = mdl(gb)["mdl_out"].float().cpu()
    scale = float(want.abs().max())
    e0, e1 = float((plain - want).abs().max()) / scale, float((got - want).abs().max()) / scale
    print(f"{sf_name}: eval logits vs the fp32 oracle {e0:.3e} -> {e1:.3e} with calibrated shifts ({n} convolutions)")
    assert torch.equal(got, again)
    assert e1 < max(0.9 * e0, 1.5e-3), (e0, e1)  # small nets at 64^2 have little weight-rounding error to remove
    mdl.sf_mdl.reset_weight_rounding()
    with torch.no_grad():
        assert torch.equal(mdl(gb)["mdl_out"].float().cpu(), plain)
        mdl.calibrate_weight_rounding({k: v.to(dev) for k, v in cal.items()})
        with torch.no_grad():
            next(iter(mdl.sf_mdl.parameters())).mul_(1.0)  # any in-place weight update bumps the version counters
        out = mdl(gb)["mdl_out"].float().cpu()
    assert all(b.wround_bias is None for b in mdl.sf_mdl._bns()), "a weight update must drop the correction"
    assert torch.equal(out, plain)


def test_grouped_weight_gradients_of_a_resblock(dev):
    """`ResBlock.group_wgrads` (round 5, vs_conv_wgrad_group): the weight gradients of a wide, few-position block as ONE
    launch of deep-pipeline blocks.  A SlowFast-R50 train-mode pass at 112^2: features bit for bit, every parameter
    gradient within fp32 summation-order distance of the per-unit launches (another position split = another order),
    grouped launches actually issued (fewer launches in the backward pass), and the grouped pass bit for bit from run to run."""
    from oracle.slowfast_ref import default_sf_cfg
    from vidsitu_amd import ops, trunk as T

    torch.manual_seed(3)
    cfg = default_sf_cfg("slowfast", 50, 64, 32)
    mdl = T.VideoTrunk(cfg).to(dev).train()
    g = torch.Generator().manual_seed(4)
    fast = torch.randn(2, 3, 32, 112, 112, generator=g).to(dev)
    xs = [fast[:, :, ::4].contiguous(), fast]
    bufs = {k: v.clone() for k, v in mdl.named_buffers()}
    lib = ops._lib.load()

    def run(group):
        T.ResBlock.group_wgrads = group
        for k, v in mdl.named_buffers():
            v.copy_(bufs[k])
        for p in mdl.parameters():
            p.grad = None
        feats = mdl.forward_features([x.clone() for x in xs])
        gg = torch.Generator().manual_seed(5)
        loss = sum((f.float() * torch.randn(f.shape, generator=gg).to(dev)).sum() for f in feats)
        n0 = lib.vs_launch_count()
        loss.backward()
        torch.cuda.synchronize()
        return ([f.detach().clone() for f in feats], {k: p.grad.clone() for k, p in mdl.named_parameters()},
                lib.vs_launch_count() - n0)

    try:
        run(True)
        f0, g0, n_sep = run(False)
        f1, g1, n_grp = run(True)
        f2, g2, _ = run(True)
    finally:
        T.ResBlock.group_wgrads = True
    assert n_grp <= n_sep - 20, (n_sep, n_grp)
    assert all(torch.equal(a, b) for a, b in zip(f0, f1))
    assert not [k for k in g1 if not torch.equal(g1[k], g2[k])], "the grouped pass must be bitwise reproducible"
    worst = max((float((g1[k] - g0[k]).abs().max() / g0[k].abs().max().clamp_min(1e-30)), k) for k in g0)
    print(f"grouped vs per-unit weight gradients: worst max-normalised difference {worst[0]:.2e} ({worst[1]}); "
          f"backward launches {n_sep} -> {n_grp}")
    assert worst[0] < 2e-5, worst
