"""The fairseq-style TransformerDecoder (`TxDecoderReal`) through the C-ABI against
`oracle/txdec_ref.py` (parity unpinned: fairseq is absent; the oracle is cross-checked against
torch.nn.TransformerDecoderLayer on the CPU): logits, every parameter gradient and the encoder-output
gradient, cached decoding == whole-sequence pass, beam search token ids bit-exact, plugin surface.
Tolerances: fp32 kernels with another summation order -- 2e-4 of the tensor's max (5e-4 for gradients,
which pass through three layers of atomically ordered / re-associated sums)."""
import numpy as np
import pytest
import torch

from oracle import beam_ref, txdec_ref

pytestmark = pytest.mark.gpu

VOC, D, FFN, NL, OUT, PAD, HEADS = 97, 128, 256, 2, 64, 96, 8  # head dim 16


def _model(dev, seed=3, d=D, heads=HEADS, ffn=FFN, out=OUT):
    from vidsitu_amd.fseq_txdec import TransformerDecoderHip

    w = txdec_ref.make_weights(VOC, d, ffn, NL, out, PAD, seed=seed)
    m = TransformerDecoderHip(VOC, d, ffn, heads, NL, out, PAD, dropout=0.0, max_positions=64)
    missing, unexpected = m.load_state_dict(w, strict=True)
    return w, m.to(dev)


def _tokens(seed=0, rows=5, l=9):
    g = torch.Generator().manual_seed(seed)
    t = torch.randint(0, VOC - 1, (rows, l), generator=g)
    t[0, 6:] = PAD
    t[3, 4:] = PAD
    return t


@pytest.mark.parametrize("with_enc", [True, False], ids=["enc", "noenc"])
@pytest.mark.parametrize("d,heads", [(128, 8), (256, 2)], ids=["dh16", "dh128"])
def test_logits_match_oracle(d, heads, with_enc, dev):
    w, m = _model(dev, d=d, heads=heads)
    toks = _tokens()
    enc = torch.randn(1, toks.shape[0], d, generator=torch.Generator().manual_seed(1)) if with_enc else None
    want = txdec_ref.forward(w, toks, enc, PAD, heads, NL)
    got = m.eval().forward_logits(toks.to(dev), None if enc is None else enc[0].to(dev)).cpu()
    valid = toks.ne(PAD)
    err = float((got - want)[valid].abs().max()) / float(want.abs().max())
    assert err < 2e-4, err


@pytest.mark.parametrize("with_enc", [True, False], ids=["enc", "noenc"])
def test_gradients_match_oracle_autograd(with_enc, dev):
    from vidsitu_amd.fseq_txdec import _TxDecTrainFn
    from vidsitu_amd.hf_gpt2_fseq import lm_loss

    w, m = _model(dev)
    toks = _tokens(seed=2)
    enc = torch.randn(1, toks.shape[0], D, generator=torch.Generator().manual_seed(5)) if with_enc else None
    # oracle: autograd through the restatement
    wg = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    enc_g = None if enc is None else enc.clone().requires_grad_(True)
    loss_ref = txdec_ref.lm_loss(txdec_ref.forward(wg, toks, enc_g, PAD, HEADS, NL), toks, PAD)
    loss_ref.backward()
    # HIP: one autograd node + the fused cross-entropy
    m.train()
    enc_d = None if enc is None else enc[0].to(dev).requires_grad_(True)
    tick = torch.zeros(1, device=dev, requires_grad=True)
    logits = _TxDecTrainFn.apply(m, toks.to(dev), enc_d, tick)
    loss = lm_loss(logits, toks.to(dev), PAD)
    assert abs(float(loss) - float(loss_ref)) < 1e-4 * max(1.0, abs(float(loss_ref)))
    loss.backward()
    worst = 0.0
    # k_proj.bias has an exactly-zero true gradient (a constant added to every key shifts all scores of a
    # query equally): both sides hold rounding noise there, so errors are measured against at least
    # 1e-3 of the largest gradient in the model
    floor = 1e-3 * max(float(v.grad.abs().max()) for v in wg.values() if v.grad is not None)
    for name in m._names:
        g_ref, g = wg[name].grad, m.P(name).grad
        if g_ref is None:  # encoder attention unused without an encoder output
            assert float(g.abs().max()) == 0.0, name
            continue
        scale = max(float(g_ref.abs().max()), floor)
        err = float((g.cpu() - g_ref).abs().max()) / scale
        worst = max(worst, err)
        assert err < 5e-4, (name, err)
    if with_enc:
        err = float((enc_d.grad.cpu() - enc_g.grad[0]).abs().max()) / float(enc_g.grad.abs().max())
        assert err < 5e-4, err
    print(f"worst relative gradient error {worst:.2e}")


@pytest.mark.parametrize("d,heads", [(128, 8), (256, 2)], ids=["dh16", "dh128"])
def test_cached_decode_equals_whole_sequence_pass(d, heads, dev):
    from vidsitu_amd.hf_gpt2_fseq import KVCacheState

    w, m = _model(dev, d=d, heads=heads)
    m.eval()
    toks = torch.randint(0, VOC - 1, (6, 10), generator=torch.Generator().manual_seed(4)).to(dev)
    enc = torch.randn(6, d, generator=torch.Generator().manual_seed(6)).to(dev)
    full = m.forward_logits(toks, enc)
    st = KVCacheState()
    m.begin_incremental(st, enc, 6, 12)
    for t in range(toks.shape[1]):
        step = m.forward_step(toks[:, t].contiguous(), st)
        assert float((step - full[:, t]).abs().max()) < 2e-4 * float(full.abs().max()), t


class _Tok:
    def __init__(self):
        self.pad_token_id, self.eos_token_id = PAD, VOC - 2

    def __len__(self):
        return VOC

    def pad(self):
        return PAD

    def eos(self):
        return VOC - 2

    def unk(self):
        return VOC - 2


class _EncDec(torch.nn.Module):
    """Minimal `SeqGenCustom` model: a fixed one-position encoder output per sentence + the decoder."""

    def __init__(self, dec_model, enc):
        super().__init__()
        from vidsitu_amd.fseq_txdec import TxDecoderReal
        from vidsitu_amd.mdl_sf_base import EncoderOut, Reorderer

        self.use_encoder = True
        dec = TxDecoderReal.__new__(TxDecoderReal)
        torch.nn.Module.__init__(dec)
        dec.model, dec.pad_idx = dec_model, PAD
        self.decoder = dec
        self._enc, self._EO, self._re = enc, EncoderOut, Reorderer()

    def max_decoder_positions(self):
        return self.decoder.model.max_positions - 1

    def forward_encoder(self, inp):
        return self._EO(encoder_out=self._enc.unsqueeze(0), encoder_padding_mask=None, encoder_embedding=None,
                        encoder_states=None, src_tokens=None, src_lengths=None)

    def reorder_encoder_out(self, encoder_out, new_order):
        return self._re.reorder_encoder_out(encoder_out, new_order)


@pytest.mark.parametrize("device_search", [True, False], ids=["device", "host"])
def test_beam_search_tokens_bit_exact_vs_oracle(device_search, dev):
    from vidsitu_amd.seq_gen import SeqGenCustom

    w, m = _model(dev, seed=9)
    eos = VOC - 2
    w = dict(w)
    w["output_projection.weight"] = w["output_projection.weight"].clone()
    w["output_projection.weight"][eos] *= 2.5  # hypotheses should finish at different steps
    with torch.no_grad():
        m.P("output_projection.weight")[eos] *= 2.5
    m.eval()
    bsz, beam = 4, 3
    enc = torch.randn(bsz, D, generator=torch.Generator().manual_seed(8))
    prefix = np.array([[5], [9], [5], [70]], dtype=np.int64)

    def step_logits(tokens, sent_ids):
        e = enc[torch.as_tensor(sent_ids)].unsqueeze(0)
        return txdec_ref.forward(w, torch.from_numpy(tokens), e, PAD, HEADS, NL)[:, -1, :].numpy()

    want = beam_ref.generate(step_logits, bsz=bsz, vocab=VOC, pad=PAD, eos=eos, unk=eos, beam_size=beam,
                             max_len_b=9, min_len=1, prefix_tokens=prefix, max_decoder_positions=63)
    lm = _EncDec(m, enc.to(dev))
    sample = {"src_tokens": torch.zeros(bsz, 1, dtype=torch.long, device=dev),
              "src_lengths": torch.ones(bsz, dtype=torch.long, device=dev)}
    for use in range(3 if device_search else 1):  # eager, graph capture, graph replay
        gen = SeqGenCustom([lm], _Tok(), beam_size=beam, max_len_b=9, min_len=1, device_search=device_search)
        got = gen._generate(sample, prefix_tokens=torch.from_numpy(prefix).to(dev))
        for sent in range(bsz):
            assert [h["tokens"].tolist() for h in got[sent]] == [h["tokens"].tolist() for h in want[sent]], (use, sent)
            for hg, hw in zip(got[sent], want[sent]):
                assert abs(float(hg["score"]) - hw["score"]) < 1e-4


def test_txdec_plugin_surface_trains(dev):
    """`get_mdl_loss_eval` row `sfpret_txe_txd_vbarg` with the reference's default `tx_dec_type: txdec`:
    loss forward + backward through decoder, TxEncoder and the feature MLP, Adam steps, generation."""
    from vidsitu_amd import synth_data
    from vidsitu_amd.extended_config import get_cfg
    from vidsitu_amd.mdl_selector import get_mdl_loss_eval
    from vidsitu_amd.optim import ArenaAdam, ParamArena

    cfg = get_cfg({"task_type": "vb_arg", "mdl.mdl_name": "sfpret_txe_txd_vbarg", "mdl.tx_dec_type": "txdec",
                   "tx_dec.decoder_layers": 2, "tx_dec.encoder_layers": 1, "synth.gpt2_vocab": 211,
                   "gen.beam_size": 2, "gen.max_len_b": 8})
    comm = synth_data.make_comm(cfg)
    sel = get_mdl_loss_eval(cfg)
    torch.manual_seed(0)
    mdl = sel["mdl"](cfg=cfg, comm=comm).to(dev).train()
    batch = synth_data.synth_srl_batch(comm, bs=2, n_ev=3, seq_len=12, device=dev)
    arena = ParamArena(mdl, adopt_conv=False)
    opt = ArenaAdam(arena, lr=3e-4)
    loss_fn = sel["loss"](cfg, comm)
    losses = []
    for _ in range(6):
        opt.zero_grad()
        loss = loss_fn(mdl(batch), batch)["loss"]
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0] - 0.2, losses
    for n, p in mdl.named_parameters():  # every parameter on the path received a gradient
        assert p.grad is not None and torch.isfinite(p.grad).all(), n
    mdl.eval()
    out = sel["evl"](cfg, comm, dev).forward_one_batch(mdl, batch)
    assert len(out) == 2 and all(len(r["vb_output"]) == 3 for r in out)


def test_fseq_encoder_matches_oracle_forward_and_backward(dev):
    """`TxEncoderOld` (fairseq TransformerEncoder over the per-event video features, `tx_enc_type: old`)."""
    from types import SimpleNamespace
    from vidsitu_amd.fseq_txdec import TxEncoderOld

    d, ffn, heads, nl, pad = 128, 256, 8, 2, 96
    tok = SimpleNamespace(pad_token_id=pad, __len__=lambda: 97)

    class _T:
        pad_token_id = pad

        def __len__(self):
            return 97

    cfg = SimpleNamespace(tx_dec=SimpleNamespace(encoder_embed_dim=d, encoder_ffn_embed_dim=ffn,
                                                 encoder_attention_heads=heads, encoder_layers=nl, dropout=0.0))
    enc = TxEncoderOld(cfg, SimpleNamespace(gpt2_hf_tok=_T()))
    w = txdec_ref.make_encoder_weights(d, ffn, nl, seed=11)
    missing, unexpected = enc.load_state_dict(w, strict=False)
    assert not unexpected and missing == ["embed_tokens.weight"]
    enc = enc.to(dev).train()
    emb = torch.randn(3, 5, d, generator=torch.Generator().manual_seed(2))
    # oracle with autograd
    wg = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    emb_ref = emb.clone().requires_grad_(True)
    out_ref = txdec_ref.encoder_forward(wg, emb_ref, emb_ref[..., 0].detach(), pad, heads, nl)
    g_out = torch.randn(out_ref.shape, generator=torch.Generator().manual_seed(3))
    (out_ref * g_out).sum().backward()
    # HIP
    emb_d = emb.to(dev).requires_grad_(True)
    out = enc(src_tokens=emb_d[..., 0].detach(), token_embeddings=emb_d, return_all_hiddens=True)
    assert tuple(out.encoder_out.shape) == (5, 3, d) and len(out.encoder_states) == nl
    err = float((out.encoder_out.detach().cpu() - out_ref.detach()).abs().max()) / float(out_ref.abs().max())
    assert err < 2e-4, err
    (out.encoder_out * g_out.to(dev)).sum().backward()
    sd = dict(enc.named_parameters())
    floor = 1e-3 * max(float(v.grad.abs().max()) for v in wg.values())
    for name, ref in wg.items():
        g = sd[name].grad.cpu()
        e = float((g - ref.grad).abs().max()) / max(float(ref.grad.abs().max()), floor)
        assert e < 5e-4, (name, e)
    e = float((emb_d.grad.cpu() - emb_ref.grad).abs().max()) / float(emb_ref.grad.abs().max())
    assert e < 5e-4, e


def test_conc_encoder_matches_oracle_forward_and_backward(dev):
    """`TxEncoderNew_Conc` (`tx_enc_type: new_conc`, mdl_sf_base.py:395-420): encoder output concatenated with
    the features and mixed by a 2-layer MLP -- output, feature gradient and every parameter gradient."""
    from types import SimpleNamespace
    from vidsitu_amd.mdl_sf_base import TxEncoderNew_Conc

    d, ffn, heads, nl, pad = 128, 256, 8, 2, 96

    class _T:
        pad_token_id = pad

        def __len__(self):
            return 97

    cfg = SimpleNamespace(tx_dec=SimpleNamespace(encoder_embed_dim=d, encoder_ffn_embed_dim=ffn,
                                                 encoder_attention_heads=heads, encoder_layers=nl, dropout=0.0))
    enc = TxEncoderNew_Conc(cfg, SimpleNamespace(gpt2_hf_tok=_T()))
    w = txdec_ref.make_encoder_weights(d, ffn, nl, seed=13)
    g = torch.Generator().manual_seed(5)
    w.update({"orig_tx_out_comb.0.weight": torch.randn(d, 2 * d, generator=g) / (2 * d) ** 0.5,
              "orig_tx_out_comb.0.bias": torch.randn(d, generator=g) * 0.1,
              "orig_tx_out_comb.2.weight": torch.randn(d, d, generator=g) / d ** 0.5,
              "orig_tx_out_comb.2.bias": torch.randn(d, generator=g) * 0.1})
    missing, unexpected = enc.load_state_dict(w, strict=False)
    assert not unexpected and missing == ["embed_tokens.weight"]
    enc = enc.to(dev).train()
    emb = torch.randn(4, 5, d, generator=torch.Generator().manual_seed(2))
    wg = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    emb_ref = emb.clone().requires_grad_(True)
    out_ref = txdec_ref.encoder_conc_forward(wg, emb_ref, emb_ref[..., 0].detach(), pad, heads, nl)
    g_out = torch.randn(out_ref.shape, generator=torch.Generator().manual_seed(3))
    (out_ref * g_out).sum().backward()
    emb_d = emb.to(dev).requires_grad_(True)
    out = enc(src_tokens=emb_d[..., 0].detach(), token_embeddings=emb_d)
    assert tuple(out.encoder_out.shape) == (5, 4, d) and out.encoder_padding_mask is None
    err = float((out.encoder_out.detach().cpu() - out_ref.detach()).abs().max()) / float(out_ref.abs().max())
    assert err < 2e-4, err
    (out.encoder_out * g_out.to(dev)).sum().backward()
    sd = dict(enc.named_parameters())
    floor = 1e-3 * max(float(v.grad.abs().max()) for v in wg.values())
    for name, ref in wg.items():
        gr = sd[name].grad.cpu()
        e = float((gr - ref.grad).abs().max()) / max(float(ref.grad.abs().max()), floor)
        assert e < 5e-4, (name, e)
    e = float((emb_d.grad.cpu() - emb_ref.grad).abs().max()) / float(emb_ref.grad.abs().max())
    assert e < 5e-4, e


@pytest.mark.parametrize("dec", ["txdec", "gpt2"])
def test_sfpret_txed_vbarg_row_trains_and_generates(dec, dev):
    """Selector row `sfpret_txed_vbarg` (`SFPreFeats_TxDec`: feature MLP -> decoder, no transformer encoder)."""
    from vidsitu_amd import synth_data
    from vidsitu_amd.extended_config import get_cfg
    from vidsitu_amd.mdl_selector import get_mdl_loss_eval
    from vidsitu_amd.optim import ArenaAdam, ParamArena

    over = {"task_type": "vb_arg", "mdl.mdl_name": "sfpret_txed_vbarg", "mdl.tx_dec_type": dec,
            "tx_dec.decoder_layers": 1, "synth.gpt2_vocab": 211, "gen.beam_size": 2, "gen.max_len_b": 6}
    if dec == "gpt2":
        over["mdl.gpt2_mdl_name"] = "gpt2-synth-tiny"
    cfg = get_cfg(over)
    comm = synth_data.make_comm(cfg)
    sel = get_mdl_loss_eval(cfg)
    torch.manual_seed(0)
    mdl = sel["mdl"](cfg=cfg, comm=comm).to(dev).train()
    batch = synth_data.synth_srl_batch(comm, bs=2, n_ev=5, seq_len=10, device=dev)
    arena = ParamArena(mdl, adopt_conv=False)
    opt = ArenaAdam(arena, lr=3e-4)
    loss_fn = sel["loss"](cfg, comm)
    losses = []
    for _ in range(5):
        opt.zero_grad()
        loss = loss_fn(mdl(batch), batch)["loss"]
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    mdl.eval()
    out = sel["evl"](cfg, comm, dev).forward_one_batch(mdl, batch)
    assert len(out) == 2 and all(len(r["vb_output"]) == 5 for r in out)


@pytest.mark.parametrize("enc_type", ["old", "new_conc"])
def test_old_encoder_plus_txdec_plugin_surface_trains(enc_type, dev):
    """The paper's SF+TxE+TxD row with both fairseq-style halves: `tx_enc_type: old` / `new_conc`,
    `tx_dec_type: txdec`."""
    from vidsitu_amd import synth_data
    from vidsitu_amd.extended_config import get_cfg
    from vidsitu_amd.mdl_selector import get_mdl_loss_eval
    from vidsitu_amd.optim import ArenaAdam, ParamArena

    cfg = get_cfg({"task_type": "vb_arg", "mdl.mdl_name": "sfpret_txe_txd_vbarg", "mdl.tx_dec_type": "txdec",
                   "mdl.tx_enc_type": enc_type, "tx_dec.decoder_layers": 1, "tx_dec.encoder_layers": 2,
                   "synth.gpt2_vocab": 211})
    comm = synth_data.make_comm(cfg)
    sel = get_mdl_loss_eval(cfg)
    torch.manual_seed(0)
    mdl = sel["mdl"](cfg=cfg, comm=comm).to(dev).train()
    batch = synth_data.synth_srl_batch(comm, bs=2, n_ev=5, seq_len=10, device=dev)
    arena = ParamArena(mdl, adopt_conv=False)
    opt = ArenaAdam(arena, lr=3e-4)
    loss_fn = sel["loss"](cfg, comm)
    losses = []
    for _ in range(6):
        opt.zero_grad()
        loss = loss_fn(mdl(batch), batch)["loss"]
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0] - 0.2, losses
