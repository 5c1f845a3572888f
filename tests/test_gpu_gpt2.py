"""Parity of the GPT-2 decoder path through the C-ABI:
fp32 GEMM / attention / embedding kernels against the oracle and the huggingface golden logits,
cached decoding against the whole-sequence pass, and the beam search (`vidsitu_amd.seq_gen`)
against the numpy restatement driven by the oracle language model -- token ids bit-exact.
Tolerances: fp32 kernels with a different summation order: 2e-4 of the tensor's max; logits
vs the golden (BASELINE: "logits within 1e-3 of reference"): 1e-3 absolute."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import beam_ref, gpt2_ref

pytestmark = pytest.mark.gpu
GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "gpt2_*.npz")))


def _model_from_golden(path, dev):
    from vidsitu_amd.hf_gpt2_fseq import GPT2LMHeadModelHip

    z = np.load(path)
    vocab, n_pos, d, n_layer, n_head, seed = [int(v) for v in z["dims"]]
    w = gpt2_ref.make_weights(vocab, n_pos, d, n_layer, seed)
    m = GPT2LMHeadModelHip(n_layer, d, n_head, n_pos, vocab)
    sd = {k: torch.from_numpy(v) for k, v in w.items()}
    sd["lm_head.weight"] = sd["transformer.wte.weight"]
    m.load_state_dict(sd, strict=True)
    return z, w, n_head, m.to(dev).eval()


@pytest.mark.parametrize("m,n,k,act,use_res", [(5, 7, 12, 0, False), (40, 1024, 1024, 2, True),
                                               (64, 96, 50, 1, False), (65, 130, 36, 2, True),
                                               (300, 3072, 1024, 0, False), (600, 1024, 4096, 0, True),
                                               (257, 97, 64, 2, False),
                                               # 17..64 rows, K % 128 == 0: the skinny MFMA kernels (three tile shapes)
                                               (17, 1024, 1024, 0, False), (33, 100, 256, 1, True),
                                               (50, 3072, 1024, 2, True), (50, 1024, 4096, 0, True),
                                               (64, 50259, 1024, 0, False), (48, 1030, 128, 2, False),
                                               (20, 24, 384, 0, True), (40, 1000, 512, 1, False),
                                               (30, 4096, 1024, 0, False), (50, 1024, 1024, 0, True),
                                               # > 64 rows with few output tiles: split-K slabs + reduce
                                               (600, 1024, 1024, 2, True), (600, 3072, 1024, 0, False),
                                               (599, 1000, 4096, 1, True), (130, 64, 520, 0, False),
                                               (1024, 600, 600, 0, False)])
def test_gemm_nt_f32(m, n, k, act, use_res, dev):
    from vidsitu_amd import ops

    g = torch.Generator().manual_seed(m * 7 + n)
    x = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g) / k ** 0.5
    b = torch.randn(n, generator=g)
    res = torch.randn(m, n, generator=g) if use_res else None
    ref = x.double() @ w.double().t() + b.double()
    if act == 1:
        ref = ref.clamp_min(0)
    elif act == 2:
        ref = torch.from_numpy(gpt2_ref.gelu_new(ref.numpy()))
    if use_res:
        ref = ref + res.double()
    args = (x.to(dev), w.to(dev), b.to(dev), res.to(dev) if use_res else None, act)
    y = ops.gemm_nt(*args)
    err = float((y.cpu().double() - ref).abs().max()) / float(ref.abs().max())
    assert err < 2e-5, f"gemm_nt {m}x{n}x{k} act {act}: rel err {err:.2e}"
    assert torch.equal(ops.gemm_nt(*args), y)


@pytest.mark.parametrize("m,n,k,act,use_res,ypk", [(50, 3072, 1024, 0, False, False), (50, 1024, 1024, 0, True, False),
                                                   (50, 4096, 1024, 2, False, True), (50, 1024, 4096, 0, True, False),
                                                   (40, 50259, 1024, 0, False, False), (17, 48, 128, 1, True, True),
                                                   (9, 1000, 256, 0, False, False), (64, 208, 512, 2, True, True),
                                                   (33, 4096, 2048, 0, False, False)])
def test_gemm_nt_packed_equals_row_major(m, n, k, act, use_res, ypk, dev):
    """Fragment-major operands (vs_pack_rows_f32) give the bits of the row-major skinny kernel -- same
    MFMA sequence, only the addresses differ -- and the round trip pack -> unpack is the identity."""
    from vidsitu_amd import ops

    g = torch.Generator().manual_seed(m + n + k)
    x = torch.randn(m, k, generator=g).to(dev)
    w = (torch.randn(n, k, generator=g) / k ** 0.5).to(dev)
    b = torch.randn(n, generator=g).to(dev)
    res = torch.randn(m, n, generator=g).to(dev) if use_res else None
    xp, wp = ops.pack_rows_f32(x), ops.pack_rows_f32(w)
    assert torch.equal(ops.unpack_rows_f32(xp, m, k), x) and torch.equal(ops.unpack_rows_f32(wp, n, k), w)
    y = ops.gemm_nt_packed(xp, wp, m, n, k, b=b, res=res, act=act, y_packed=ypk)
    if ypk:
        y = ops.unpack_rows_f32(y, m, n)
    ref = x.double() @ w.double().t() + b.double()
    if act == 1:
        ref = ref.clamp_min(0)
    elif act == 2:
        ref = torch.from_numpy(gpt2_ref.gelu_new(ref.cpu().numpy())).to(dev)
    if use_res:
        ref = ref + res.double()
    err = float((y.double() - ref).abs().max()) / float(ref.abs().max())
    assert err < 2e-5, f"packed gemm {m}x{n}x{k}: rel err {err:.2e}"
    if m > 16:
        assert torch.equal(y, ops.gemm_nt(x, w, b, res, act))


def test_packed_layernorm_and_attention_outputs(dev):
    from vidsitu_amd import ops

    g = torch.Generator().manual_seed(5)
    rows, d, heads, lmax, t = 37, 1024, 16, 9, 6
    x = torch.randn(rows, d, generator=g).to(dev)
    gamma, beta = torch.randn(d, generator=g).to(dev), torch.randn(d, generator=g).to(dev)
    want = ops.add_layernorm_fwd(x, None, gamma, beta, 1e-5)[0]
    got = ops.unpack_rows_f32(ops.layernorm_fwd_packed(x, gamma, beta, 1e-5), rows, d)
    assert torch.equal(got, want)
    qkv = torch.randn(rows, 3 * d, generator=g).to(dev)
    kc = torch.randn(rows, heads, lmax, d // heads, generator=g).to(dev)
    vc = torch.randn(rows, heads, lmax, d // heads, generator=g).to(dev)
    anc = torch.randint(0, rows, (rows, lmax), generator=g).to(torch.int32).to(dev)
    o_row = ops.attn_decode(qkv, kc.clone(), vc.clone(), None, t, ancestry=anc)
    o_pk = ops.attn_decode(qkv, kc.clone(), vc.clone(), None, t, ancestry=anc, out_packed=True)
    assert torch.equal(ops.unpack_rows_f32(o_pk, rows, d), o_row)


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_gpt2_logits_match_transformers_golden(path, dev):
    z, w, n_head, m = _model_from_golden(path, dev)
    toks, mask = torch.from_numpy(z["tokens"]).to(dev), torch.from_numpy(z["mask"]).to(dev)
    logits = m.forward_logits(toks, mask).cpu().numpy()
    valid = z["mask"].astype(bool)
    if "logits" in z:
        err = np.abs(logits - z["logits"])[valid].max()
    else:
        err = np.abs(logits[:, :, :64] - z["logits_first64"])[valid].max()
        assert (logits.argmax(-1) == z["logits_argmax"])[valid].all()
        assert np.abs(logits.max(-1) - z["logits_max"])[valid].max() < 1e-3
    print(f"{os.path.basename(path)}: max |logit - golden| = {err:.3e}")
    assert err < 1e-3


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_cached_decode_equals_whole_sequence_pass(path, dev):
    from vidsitu_amd.hf_gpt2_fseq import KVCacheState

    z, w, n_head, m = _model_from_golden(path, dev)
    toks = torch.from_numpy(z["tokens"]).to(dev)  # no padding in generation
    toks = toks.clamp(min=1)[:, :12].contiguous()
    if "medium" in path:  # 20 rows: the fragment-major (packed) decode step
        toks = torch.cat([toks, toks.flip(1), (toks * 7 + 1) % 50000] * 3)[:20].contiguous()
    full = m.forward_logits(toks, None)
    st = KVCacheState()
    for t in range(toks.shape[1]):
        step = m.forward_step(toks[:, t].contiguous(), st, max_len=toks.shape[1])
        assert float((step - full[:, t]).abs().max()) < 2e-4 * float(full.abs().max())


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_ancestry_table_equals_cache_gather(path, dev):
    """A beam reorder through the ancestry table (vs_attn_decode reads position j of row r from cache
    row anc[r][j]) gives the bits of the physical gather (vs_kv_gather)."""
    from vidsitu_amd.hf_gpt2_fseq import KVCacheState

    z, w, n_head, m = _model_from_golden(path, dev)
    rows, steps = 6, 7
    g = torch.Generator().manual_seed(3)
    toks = torch.randint(1, 60, (steps, rows), generator=g).to(dev)
    perms = [torch.randint(0, rows, (rows,), generator=g).to(dev) for _ in range(steps)]
    a, b = KVCacheState(), KVCacheState()
    b.anc = torch.arange(rows, dtype=torch.int32, device=dev).view(-1, 1).repeat(1, steps).contiguous()
    for t in range(steps):
        if t > 0:  # new row r continues old row perm[r]
            m.reorder_state(a, perms[t])
            anc = b.anc[perms[t]].contiguous()
            anc[:, t:] = torch.arange(rows, dtype=torch.int32, device=dev).view(-1, 1)
            b.anc = anc
        la = m.forward_step(toks[t].contiguous(), a, max_len=steps)
        lb = m.forward_step(toks[t].contiguous(), b, max_len=steps)
        assert torch.equal(la, lb), f"step {t}"


def test_lm_loss_with_ignore_index(dev):
    from vidsitu_amd.hf_gpt2_fseq import lm_loss

    z, w, n_head, m = _model_from_golden(GOLD[1], dev)
    pad = int(z["pad"])
    toks, mask = torch.from_numpy(z["tokens"]).to(dev), torch.from_numpy(z["mask"]).to(dev)
    logits = m.forward_logits(toks, mask)
    got = float(lm_loss(logits, toks, pad))
    want = gpt2_ref.lm_loss(gpt2_ref.forward(w, z["tokens"], z["mask"], n_head), z["tokens"], pad)
    assert abs(got - want) < 1e-4 * max(1.0, abs(want)), (got, want)


@pytest.mark.parametrize("V", [1000, 5000, 50259])  # one block per row / 3 and 25 slices per row
def test_beam_topk_kernel_rules(V, dev):
    from vidsitu_amd import ops

    rs = np.random.RandomState(0)
    rows, k = 6, 10
    pad, eos, unk = V - 1, V - 2, 5
    x = rs.randn(rows, V).astype(np.float32) * 3
    x[0, 17] = x[0, 400]            # an exact tie: lowest token first
    x[4, V - 3] = x[4, 2] = 30.0    # a tie across slices
    x[1, 3] = np.nan                # NaN -> -inf
    cum = rs.randn(rows).astype(np.float32)
    cum[5] = -np.inf                # a dead beam: all -inf, indices 0,1,2,...
    forced = np.array([-1, -1, 42, pad, -1, -1], dtype=np.int64)
    for flags in (0, 1, 2):
        lp = beam_ref.log_softmax(x / np.float32(0.7))
        lp[lp != lp] = -np.inf
        lp[:, pad] = -np.inf
        lp[:, unk] -= 0.25
        if flags & 1:
            lp[:, :eos] = -np.inf
            lp[:, eos + 1:] = -np.inf
        for r in range(rows):
            if forced[r] >= 0 and forced[r] != pad:
                keep = lp[r, forced[r]]
                lp[r] = -np.inf
                lp[r, forced[r]] = keep
            elif flags & 2:
                lp[r, eos] = -np.inf
        want_v, want_i = beam_ref.topk_lowest_index(lp + cum[:, None], k)
        v, i = ops.beam_topk(torch.from_numpy(x).to(dev), torch.from_numpy(cum).to(dev),
                             torch.from_numpy(forced).to(dev), k, pad, eos, unk, unk_penalty=0.25,
                             temperature=0.7, eos_only=bool(flags & 1), ban_eos=bool(flags & 2))
        assert np.array_equal(i.cpu().numpy(), want_i), f"flags {flags}"
        fin = np.isfinite(want_v)
        assert np.allclose(v.cpu().numpy()[fin], want_v[fin], atol=2e-5)
        assert np.array_equal(np.isneginf(v.cpu().numpy()), np.isneginf(want_v))


class _Tok:
    def __init__(self, vocab, pad, eos, unk):
        self.v, self._pad, self._eos, self._unk = vocab, pad, eos, unk
        self.pad_token_id, self.eos_token_id = pad, eos

    def __len__(self):
        return self.v

    def pad(self):
        return self._pad

    def eos(self):
        return self._eos

    def unk(self):
        return self._unk


class _LM(torch.nn.Module):
    """Minimal model object of the `SeqGenCustom` contract around a GPT2LMHeadModelHip."""

    def __init__(self, m, pad):
        super().__init__()
        from vidsitu_amd.hf_gpt2_fseq import HuggingFaceGPT2Decoder

        self.use_encoder = False
        dec = HuggingFaceGPT2Decoder.__new__(HuggingFaceGPT2Decoder)
        torch.nn.Module.__init__(dec)
        dec.model, dec.pad_idx = m, pad
        self.decoder = dec

    def max_decoder_positions(self):
        return self.decoder.model.config.n_positions - 1

    def forward_encoder(self, inp):
        return None


@pytest.mark.parametrize("device_search", [True, False], ids=["device", "host"])
@pytest.mark.parametrize("beam,min_len,max_len_b,use_prefix,kv", [(1, 0, 6, True, True), (3, 1, 7, True, True),
                                                                  (3, 1, 7, True, False), (4, 0, 5, False, True),
                                                                  (5, 2, 9, True, True), (2, 0, 20, True, True)])
def test_beam_search_tokens_bit_exact_vs_oracle(beam, min_len, max_len_b, use_prefix, kv, device_search, dev):
    from vidsitu_amd.seq_gen import SeqGenCustom

    z, w, n_head, m = _model_from_golden(GOLD[0], dev)
    vocab = int(z["dims"][0])
    pad, eos, unk = vocab - 1, vocab - 2, vocab - 2
    # make eos reasonably likely so that hypotheses finish at different steps
    w = dict(w)
    w["transformer.wte.weight"] = w["transformer.wte.weight"].copy()
    w["transformer.wte.weight"][eos] *= 3.0
    with torch.no_grad():
        m.P("transformer.wte.weight")[eos] *= 3.0
    bsz = 4
    prefix = np.array([[5], [9], [5], [70]], dtype=np.int64) if use_prefix else None

    def step_logits(tokens, sent_ids):
        mask = (tokens != pad).astype(np.int64)
        return gpt2_ref.forward(w, tokens, mask, n_head)[:, -1, :]

    want = beam_ref.generate(step_logits, bsz=bsz, vocab=vocab, pad=pad, eos=eos, unk=unk,
                             beam_size=beam, max_len_b=max_len_b, min_len=min_len,
                             prefix_tokens=prefix, max_decoder_positions=int(z["dims"][1]) - 1)
    gen = SeqGenCustom([_LM(m, pad)], _Tok(vocab, pad, eos, unk), beam_size=beam, max_len_b=max_len_b,
                       min_len=min_len, use_kv_cache=kv, device_search=device_search)
    sample = {"src_tokens": torch.zeros(bsz, 1, dtype=torch.long, device=dev),
              "src_lengths": torch.ones(bsz, dtype=torch.long, device=dev)}
    got = gen._generate(sample, prefix_tokens=None if prefix is None else torch.from_numpy(prefix).to(dev))
    assert len(got) == bsz
    for sent in range(bsz):
        assert len(got[sent]) == len(want[sent]) == min(beam, vocab - 1)
        for hg, hw in zip(got[sent], want[sent]):
            assert hg["tokens"].tolist() == hw["tokens"].tolist(), f"sentence {sent}"
            assert abs(float(hg["score"]) - hw["score"]) < 1e-4
            assert np.allclose(hg["positional_scores"].cpu().numpy(), hw["positional_scores"], atol=1e-4)


def test_device_search_graph_replays_equal_the_oracle(dev):
    """Uses 1 / 2 / 3+ of one generation shape run eagerly / capture one hipGraph per step / replay
    them; every use gets different prefix tokens and must reproduce the oracle's tokens."""
    from vidsitu_amd.seq_gen import SeqGenCustom

    z, w, n_head, m = _model_from_golden(GOLD[0], dev)
    vocab = int(z["dims"][0])
    pad, eos, unk = vocab - 1, vocab - 2, vocab - 2
    w = dict(w)
    w["transformer.wte.weight"] = w["transformer.wte.weight"].copy()
    w["transformer.wte.weight"][eos] *= 3.0
    with torch.no_grad():
        m.P("transformer.wte.weight")[eos] *= 3.0
    bsz, beam = 4, 3
    lm = _LM(m, pad)

    def step_logits(tokens, sent_ids):
        return gpt2_ref.forward(w, tokens, (tokens != pad).astype(np.int64), n_head)[:, -1, :]

    sample = {"src_tokens": torch.zeros(bsz, 1, dtype=torch.long, device=dev),
              "src_lengths": torch.ones(bsz, dtype=torch.long, device=dev)}
    for use in range(4):
        prefix = np.array([[5 + use], [9], [50 - use], [70]], dtype=np.int64)
        want = beam_ref.generate(step_logits, bsz=bsz, vocab=vocab, pad=pad, eos=eos, unk=unk,
                                 beam_size=beam, max_len_b=9, min_len=1, prefix_tokens=prefix,
                                 max_decoder_positions=int(z["dims"][1]) - 1)
        gen = SeqGenCustom([lm], _Tok(vocab, pad, eos, unk), beam_size=beam, max_len_b=9, min_len=1)
        got = gen._generate(sample, prefix_tokens=torch.from_numpy(prefix).to(dev))
        ses = list(lm.decoder._vs_search_sessions.values())[-1]
        assert ses.uses == use + 1 and (len(ses.graphs) > 0) == (use >= 1)
        for sent in range(bsz):
            assert [h["tokens"].tolist() for h in got[sent]] == [h["tokens"].tolist() for h in want[sent]]
            for hg, hw in zip(got[sent], want[sent]):
                assert abs(float(hg["score"]) - hw["score"]) < 1e-4


def test_device_search_equals_host_search_on_a_gpt2_medium_slice(dev):
    """d_model 1024 / 50 300 tokens / 40 rows: the skinny MFMA GEMMs with K-split, the sliced top-k, the
    dh = 64 ancestry attention and the per-step graphs against the host loop with the cache gather."""
    from vidsitu_amd.seq_gen import SeqGenCustom

    path = [p for p in GOLD if "medium" in p][0]
    z, w, n_head, m = _model_from_golden(path, dev)
    vocab = int(z["dims"][0])
    pad, eos, unk = vocab - 1, vocab - 2, vocab - 2
    bsz, beam = 8, 5
    lm = _LM(m, pad)
    sample = {"src_tokens": torch.zeros(bsz, 1, dtype=torch.long, device=dev),
              "src_lengths": torch.ones(bsz, dtype=torch.long, device=dev)}
    for use in range(3):
        prefix = torch.randint(0, 1000, (bsz, 2), generator=torch.Generator().manual_seed(use)).to(dev)
        outs = []
        for device_search in (False, True):
            gen = SeqGenCustom([lm], _Tok(vocab, pad, eos, unk), beam_size=beam, max_len_b=12, min_len=3,
                               device_search=device_search)
            outs.append(gen._generate(sample, prefix_tokens=prefix))
        for sent in range(bsz):
            assert [h["tokens"].tolist() for h in outs[0][sent]] == [h["tokens"].tolist() for h in outs[1][sent]]
            for hh, hd in zip(outs[0][sent], outs[1][sent]):
                # the host loop drops finished sentences, so its later steps run other GEMM kernels
                # (fewer rows): same tokens, scores equal to fp32 summation-order noise
                assert abs(float(hh["score"]) - float(hd["score"])) < 1e-5 * max(1.0, abs(float(hh["score"])))
                assert torch.allclose(hh["positional_scores"], hd["positional_scores"], atol=2e-5)


def test_simple_txdec_plugin_surface(dev):
    """`get_mdl_loss_eval` rows `tx_only` / `sfpret_txe_txd_vbarg` with tx_dec_type gpt2: LM loss
    forward, `forward_gen` through `EvalB_Gen` (cfg.gen), output tensor contract [B, E, 1, L]."""
    from vidsitu_amd import synth_data
    from vidsitu_amd.extended_config import get_cfg
    from vidsitu_amd.mdl_selector import get_mdl_loss_eval

    for name in ("tx_only", "sfpret_txe_txd_vbarg"):
        cfg = get_cfg({"task_type": "vb_arg", "mdl.mdl_name": name, "mdl.tx_dec_type": "gpt2",
                       "mdl.gpt2_mdl_name": "gpt2-synth-tiny", "synth.gpt2_vocab": 97,
                       "gen.beam_size": 2, "gen.max_len_b": 8})
        comm = synth_data.make_comm(cfg)
        sel = get_mdl_loss_eval(cfg)
        torch.manual_seed(0)
        mdl = sel["mdl"](cfg=cfg, comm=comm).to(dev).eval()
        batch = synth_data.synth_srl_batch(comm, bs=2, n_ev=5, seq_len=12, feat_dim=2304, device=dev)
        out = mdl(batch)
        assert tuple(out["logits"].shape) == (10, 12, 97) and torch.isfinite(out["loss"])
        want = gpt2_ref.lm_loss(out["logits"].cpu().numpy(), batch["seq_out_by_ev"].view(10, 12).cpu().numpy(),
                                comm.gpt2_hf_tok.pad_token_id)
        assert abs(float(out["loss"]) - want) < 1e-4 * max(1.0, abs(want))
        assert abs(float(sel["loss"](cfg, comm)(out, batch)["loss"]) - float(out["loss"])) == 0.0
        res = sel["evl"](cfg, comm, dev).forward_one_batch(mdl, batch)
        assert len(res) == 2 and set(res[0]["vb_output"]) == {f"Ev{i}" for i in range(1, 6)}
        for r, b in zip(res, range(2)):
            for e in range(5):
                toks = r["vb_output"][f"Ev{e + 1}"]["tokens"]
                assert toks[0] == int(batch["seq_out_by_ev"][b, e, 0, 0])  # forced first token
                assert len(toks) <= 9


@pytest.mark.parametrize("path", [p for p in GOLD if "medium" not in p], ids=lambda p: os.path.basename(p))
def test_gpt2_training_gradients_match_transformers_golden(path, dev):
    """Simple_TxDec's training math (`mdl_sf_base.py:653-667`): teacher-forced logits, shift-by-one
    CE with ignore_index = pad, loss.backward() -- every parameter gradient of the HIP backward
    against the gradients huggingface transformers computed for the same weights and tokens."""
    from vidsitu_amd.hf_gpt2_fseq import HuggingFaceGPT2Decoder, lm_loss

    z, w, n_head, m = _model_from_golden(path, dev)
    pad = int(z["pad"])
    dec = HuggingFaceGPT2Decoder.__new__(HuggingFaceGPT2Decoder)
    torch.nn.Module.__init__(dec)
    dec.model, dec.pad_idx = m, pad
    dec.train()
    toks = torch.from_numpy(z["tokens"]).to(dev)
    logits = dec(toks)[0]
    assert logits.requires_grad
    loss = lm_loss(logits, toks, pad)
    loss.backward()
    assert abs(float(loss) - float(z["loss"])) < 1e-4 * max(1.0, abs(float(z["loss"])))
    worst = 0.0
    for k in z.files:
        if not k.startswith("grad."):
            continue
        want = z[k]
        got = m.P(k[5:]).grad.cpu().numpy()
        assert got.shape == want.shape, k
        err = float(np.abs(got - want).max()) / max(float(np.abs(want).max()), 1e-12)
        worst = max(worst, err)
        assert err < 5e-4, f"{k}: relative error {err:.3e}"
    print(f"{os.path.basename(path)}: worst relative gradient error {worst:.3e}")


def test_gpt2_training_step_through_plugin_surface(dev):
    """`tx_only` model in train mode: loss -> backward -> fused Adam on the parameter arena; the
    loss of the same batch goes down."""
    from vidsitu_amd import synth_data
    from vidsitu_amd.extended_config import get_cfg
    from vidsitu_amd.mdl_selector import get_mdl_loss_eval
    from vidsitu_amd.optim import ArenaAdam, ParamArena

    cfg = get_cfg({"task_type": "vb_arg", "mdl.mdl_name": "tx_only", "mdl.tx_dec_type": "gpt2",
                   "mdl.gpt2_mdl_name": "gpt2-synth-tiny", "synth.gpt2_vocab": 97})
    comm = synth_data.make_comm(cfg)
    sel = get_mdl_loss_eval(cfg)
    torch.manual_seed(0)
    mdl = sel["mdl"](cfg=cfg, comm=comm).to(dev).train()
    loss_fn = sel["loss"](cfg, comm)
    batch = synth_data.synth_srl_batch(comm, bs=2, n_ev=5, seq_len=12, device=dev)
    arena = ParamArena(mdl, adopt_conv=False)
    opt = ArenaAdam(arena, lr=3e-3)
    losses = []
    for _ in range(6):
        opt.zero_grad()
        loss = loss_fn(mdl(batch), batch)["loss"]
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0] - 0.05, losses


def _greedy_cases():
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "greedy_gpt2_tiny.npz"))
    return z, range(int(z["n_cases"]))


@pytest.mark.parametrize("ci", list(_greedy_cases()[1]))
@pytest.mark.parametrize("sync_every", [1, 8])
def test_greedy_generation_tokens_equal_transformers_golden(ci, sync_every, dev):
    """`Simple_GPT2(_New).forward_gen`'s huggingface `generate` call (mdl_sf_base.py:494-503, 577-585):
    token ids bit-exact against tests/golden/greedy_gpt2_tiny.npz (rows stopping at eos and padded, rows
    running to max_length, a batch that stops early), whether the host tests every step or every 8th."""
    from vidsitu_amd.hf_gpt2_fseq import GPT2LMHeadModelHip

    z, _ = _greedy_cases()
    vocab, n_pos, d, n_layer, n_head, seed, max_length, pad, eos = [int(v) for v in z[f"c{ci}_dims"]]
    w = gpt2_ref.make_weights(vocab, n_pos, d, n_layer, seed)
    m = GPT2LMHeadModelHip(n_layer, d, n_head, n_pos, vocab)
    sd = {k: torch.from_numpy(v) for k, v in w.items()}
    sd["lm_head.weight"] = sd["transformer.wte.weight"]
    m.load_state_dict(sd, strict=True)
    m = m.to(dev).eval()
    first = torch.from_numpy(z[f"c{ci}_first"]).to(dev)
    got = m.generate_greedy(first, max_length, pad, eos, sync_every=sync_every).cpu().numpy()
    want = z[f"c{ci}_out"]
    assert got.shape == want.shape and (got == want).all()
    assert (gpt2_ref.greedy_generate(w, z[f"c{ci}_first"], max_length, pad, eos, n_head) == want).all()


def test_new_gpt2_only_row_trains_and_generates(dev):
    """`get_mdl_loss_eval` row `new_gpt2_only` (Simple_GPT2_New, mdl_sf_base.py:560-587): LM loss against the
    oracle, a few Adam steps lower it, `forward_gen` through EvalB_Gen returns [B, E, 1, <= 61] token ids
    equal to the oracle's greedy continuation of the trained weights."""
    from vidsitu_amd import synth_data
    from vidsitu_amd.extended_config import get_cfg
    from vidsitu_amd.mdl_selector import get_mdl_loss_eval
    from vidsitu_amd.optim import ArenaAdam, ParamArena

    cfg = get_cfg({"task_type": "vb_arg", "mdl.mdl_name": "new_gpt2_only", "mdl.gpt2_mdl_name": "gpt2-synth-tiny",
                   "synth.gpt2_vocab": 97})
    comm = synth_data.make_comm(cfg)
    sel = get_mdl_loss_eval(cfg)
    torch.manual_seed(0)
    mdl = sel["mdl"](cfg=cfg, comm=comm).to(dev)
    assert type(mdl).__name__ == "Simple_GPT2_New"
    batch = synth_data.synth_srl_batch(comm, bs=2, n_ev=5, seq_len=12, device=dev)
    pad, eos = comm.gpt2_hf_tok.pad_token_id, comm.gpt2_hf_tok.eos_token_id
    out = mdl.eval()(batch)
    want = gpt2_ref.lm_loss(out["logits"].cpu().numpy(), batch["seq_out_by_ev"].view(10, 12).cpu().numpy(), pad)
    assert abs(float(out["loss"]) - want) < 1e-4 * max(1.0, abs(want))
    mdl.train()
    loss_fn = sel["loss"](cfg, comm)
    arena = ParamArena(mdl, adopt_conv=False)
    opt = ArenaAdam(arena, lr=3e-3)
    losses = []
    for _ in range(6):
        opt.zero_grad()
        loss = loss_fn(mdl(batch), batch)["loss"]
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0] - 0.05, losses
    mdl.eval()
    sents = mdl.forward_gen(batch)
    assert sents.shape[:3] == (2, 5, 1) and sents.shape[3] <= 61
    w = {k: v.detach().cpu().numpy() for k, v in mdl.gpt2_mdl.state_dict().items() if k != "lm_head.weight"}
    first = batch["seq_out_by_ev"][:, :, 0, :1].reshape(10, 1).cpu().numpy()
    ref = gpt2_ref.greedy_generate(w, first, 61, pad, eos, 4)
    assert sents.view(10, -1).shape == ref.shape and (sents.view(10, -1).cpu().numpy() == ref).all()
    res = sel["evl"](cfg, comm, dev).forward_one_batch(mdl, batch)
    assert len(res) == 2 and set(res[0]["vb_output"]) == {f"Ev{i}" for i in range(1, 6)}


def _close(a, b):
    # the embedding backward accumulates rows with atomics: equal up to fp32 summation order
    return bool(((a - b).abs().max() <= 1e-5 * a.abs().max().clamp_min(1e-12)).item())


def test_lm_loss_backward_has_no_host_sync_and_two_forwards_keep_their_own_state(dev):
    """(1) Two forwards before the backwards: each autograd node carries its own saved activations.
    (2) `_XentIgnoreFn.backward` hands the upstream gradient to the kernel as a device pointer: it runs under
    torch's sync-debug mode "error" (a `float(go)` would raise) and equals torch's cross entropy."""
    from vidsitu_amd.hf_gpt2_fseq import _GPT2TrainFn, lm_loss

    path = [p for p in GOLD if "medium" not in p][0]
    z, _, _, m = _model_from_golden(path, dev)
    m.train()
    pad = int(z["pad"])
    for p in m.parameters():
        p.grad = torch.zeros_like(p)
    t0 = torch.from_numpy(z["tokens"]).to(dev)
    toks = [t0, torch.roll(t0, 1, dims=0).flip(1).contiguous()]
    mask = [t.ne(pad) for t in toks]

    def run(tk, mk, scale):
        tick = torch.zeros(1, device=dev, requires_grad=True)
        logits = _GPT2TrainFn.apply(m, tk, mk, tick)
        return lm_loss(logits, tk, pad) * scale

    # two forwards, then the two backwards in the opposite order == each pair alone
    want = []
    for i in range(2):
        for p in m.parameters():
            p.grad.zero_()
        run(toks[i], mask[i], 1.0 + i).backward()
        want.append([p.grad.clone() for p in m.parameters()])
    la, lb = run(toks[0], mask[0], 1.0), run(toks[1], mask[1], 2.0)
    for p in m.parameters():
        p.grad.zero_()
    lb.backward()
    got_b = [p.grad.clone() for p in m.parameters()]
    for p in m.parameters():
        p.grad.zero_()
    la.backward()
    got_a = [p.grad.clone() for p in m.parameters()]
    for w, gt in zip(want[0], got_a):
        assert _close(w, gt)
    for w, gt in zip(want[1], got_b):
        assert _close(w, gt)
    # the loss node's backward reads the upstream gradient on the device: no host synchronisation
    logits = torch.randn(6, 9, 50, device=dev, requires_grad=True)
    tk = torch.randint(0, 49, (6, 9), device=dev)
    loss = lm_loss(logits, tk, 49) * 3.0
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        loss.backward()
    finally:
        torch.cuda.set_sync_debug_mode("default")
    ref = torch.nn.functional.cross_entropy(logits.detach()[:, :-1].reshape(-1, 50).requires_grad_(),
                                            tk[:, 1:].reshape(-1), ignore_index=49)
    lg = logits.detach().clone().requires_grad_()
    (torch.nn.functional.cross_entropy(lg[:, :-1].reshape(-1, 50), tk[:, 1:].reshape(-1), ignore_index=49)
     * 3.0).backward()
    assert abs(float(ref) * 3.0 - float(loss)) < 1e-5 * abs(float(loss))
    assert _close(lg.grad, logits.grad)
