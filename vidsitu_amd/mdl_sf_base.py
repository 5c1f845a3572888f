"""Host-side mirror of `vidsitu_code/mdl_sf_base.py` for the hot path, on the HIP
kernels: `SlowFast_FeatModel` / `ResNet_FeatModel` (:20-62), `ResNetBasicHead_Trimmed`
(:65-113), `SFBase` (:116-216), `LossB` (:219-231), `LossLambda` (:234-243),
`TxEncoderNew` / `TxEncoder()` (:341-432), `GPT2_hf_fseqDec` / `TxDecoder()` (:449-464),
`Simple_TxDec` (:595-675), `Reorderer` (:694-748), `get_head_dim` (:751-760) and
`SFPreFeats_TxEncDec` (:793-832).

Same constructor contracts (`mdl(cfg=cfg, comm=comm)`, `loss(cfg, comm)`), same
attribute names (`sf_mdl`, `head`, `proj_head`, `vid_feat_encoder`,
`vid_feat_txenc`), same state_dict keys, same input dict
(`frms_ev_fast_tensor`, `frms_ev_slow_tensor`, `vseg_idx`, `label_tensor`).
The one deliberate generalisation: the number of events per video is read from
the input tensor's second axis instead of the literal 5 (`:209`), so the
BASELINE "8 clips" batch is expressible.
"""
from collections import namedtuple
from typing import Dict

import torch
from torch import nn

from . import ops
from .trunk import VideoTrunk
from .transformer_code import Transformer as TxCodeEnc, LinearFn
from .hf_gpt2_fseq import (GPT2_DIMS, GPT2LMHeadModelHip, HuggingFaceGPT2Decoder, _GPT2TrainFn,
                           lm_loss as gpt2_lm_loss)
from .fseq_txdec import TxDecoderReal, TxEncoderOld

EncoderOut = namedtuple(
    "EncoderOut",
    ["encoder_out", "encoder_padding_mask", "encoder_embedding", "encoder_states", "src_tokens",
     "src_lengths"],
)


def combine_first_ax(t, keepdim=False):
    """`utils/misc_utils.py:1-5`."""
    s = t.shape
    if keepdim:
        return t.view(1, s[0] * s[1], *s[2:])
    return t.view(s[0] * s[1], *s[2:])


class SlowFast_FeatModel(VideoTrunk):
    """mdl_sf_base.py:20-42 (two pathways)."""


class ResNet_FeatModel(VideoTrunk):
    """mdl_sf_base.py:45-62 (c2d / i3d / slow)."""


class _AvgPoolCatFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *feats):
        ctx.shapes = [tuple(f.shape) for f in feats]
        return ops.avgpool_cat(list(feats))

    @staticmethod
    def backward(ctx, dout):
        return tuple(ops.avgpool_cat_bwd(dout.float(), ctx.shapes))


class ResNetBasicHead_Trimmed(nn.Module):
    """AdaptiveAvgPool3d((1,1,1)) per pathway + channel concat (mdl_sf_base.py:65-113)."""

    def __init__(self, dim_in, pool_size):
        super().__init__()
        assert len({len(pool_size), len(dim_in)}) == 1, "pathway dimensions are not consistent."
        assert all(p is None for p in pool_size), "only the adaptive (None) pool is on the hot path"
        self.num_pathways = len(pool_size)
        self.dim_in = dim_in

    def forward(self, inputs):
        assert len(inputs) == self.num_pathways, \
            "Input tensor does not contain {} pathway".format(self.num_pathways)
        feats = [ops_act(f) for f in inputs]
        out = _AvgPoolCatFn.apply(*feats)  # [N, sum C] fp32
        return out.view(out.shape[0], out.shape[1], 1, 1, 1)


def ops_act(t):
    from .trunk import ops_ensure_act
    return ops_ensure_act(t)


class HipMLP(nn.Sequential):
    """nn.Sequential(Linear, ReLU, Linear) parameter layout (keys `0.*`, `2.*`), HIP compute."""

    def forward(self, x):
        mods = list(self)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.Linear):
                relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                x = LinearFn.apply(x, m.weight, m.bias, relu, None)
                i += 2 if relu else 1
            else:
                raise NotImplementedError(type(m))
        return x


class _XentFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels):
        loss, dlogits = ops.softmax_xent(logits, labels, want_grad=True)
        ctx.save_for_backward(dlogits)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        (dlogits,) = ctx.saved_tensors
        return dlogits * dloss, None


def hip_cross_entropy(logits, labels):
    """mean F.cross_entropy (mdl_sf_base.py:230) on vs_softmax_xent."""
    return _XentFn.apply(logits.float().contiguous(), labels)


def _first_annotation(t):
    """[B, E, n_ann, L] -> [B * E, L] of annotation 0 (the reference indexes `[:, :, [0], :]` and reshapes)."""
    b, e, _, length = t.shape
    return t[:, :, 0, :].reshape(b * e, length)


def _num_events(inp):
    for key in ("frms_ev_raw_u8", "frms_ev_fast_u8", "frms_ev_fast_tensor"):
        if key in inp:
            return inp[key].shape[1]
    raise KeyError("no frame tensor in the batch")


_TRUNK_BY_NAME = {"SlowFast": SlowFast_FeatModel, "ResNet": ResNet_FeatModel}
_PATHWAYS = {"multi": 2, "single": 1}


class SFBase(nn.Module):
    """Verb prediction from frames (`mdl_sf_base.py:116-216`): trunk -> trimmed head -> two-layer projection.
    Surface the reference and its checkpoints address: `sf_mdl`, `head`, `proj_head`, `forward_encoder`,
    `forward_decoder`, `get_feats`, and the `build_*` hooks its subclasses extend."""

    def __init__(self, cfg, comm):
        super().__init__()
        self.full_cfg, self.sf_cfg, self.cfg, self.comm = cfg, cfg.sf_mdl, cfg.mdl, comm
        self.build_model()

    def build_model(self):
        for hook in (self.build_sf_model, self.build_head, self.build_projection_head):
            hook(self.sf_cfg)

    def build_sf_model(self, cfg):
        trunk_cls = _TRUNK_BY_NAME.get(cfg.MODEL.MODEL_NAME)
        if trunk_cls is None:
            raise NotImplementedError(f"MODEL.MODEL_NAME={cfg.MODEL.MODEL_NAME}")
        self.sf_mdl = trunk_cls(cfg)

    def build_head(self, cfg):
        widths = list(self.sf_mdl.dim_out)
        n_path = _PATHWAYS.get(self.comm.path_type)
        if n_path is None:
            return  # (the reference builds no head for an unknown path type either)
        if n_path == 2:  # slow: 32 x width, fast: that over beta_inv (2048 + 256 for R50)
            full = 32 * cfg.RESNET.WIDTH_PER_GROUP
            assert widths == [full, full // cfg.SLOWFAST.BETA_INV], widths
        self.head = ResNetBasicHead_Trimmed(dim_in=widths, pool_size=[None] * n_path)

    def build_projection_head(self, cfg, out_dim=None):
        feat = sum(self.head.dim_in)
        n_out = len(self.comm.vb_id_vocab) if out_dim is None else out_dim
        self.proj_head = HipMLP(nn.Linear(feat, feat // 2), nn.ReLU(), nn.Linear(feat // 2, n_out))

    def get_feats(self, inp):
        """The trunk's input list, events folded into the batch axis: [slow, fast], [fast], or one uint8 tensor."""
        if "frms_ev_raw_u8" in inp:
            # decoded frames at their source size [B, E, T, H0, W0, 3]: the loader's
            # `img.resize((224, 224))` (dat_loader.py:188, PIL bicubic) runs on the GPU, bit-exact
            side = int(self.sf_cfg.DATA.TRAIN_CROP_SIZE)
            return [combine_first_ax(ops.resize_bicubic_u8(inp["frms_ev_raw_u8"], side, side))]
        if "frms_ev_fast_u8" in inp:
            # optional fast path beside the reference contract: the loader's uint8 RGB frames
            # [B, E, T, H, W, 3]; normalisation and the slow-pathway gather happen on the GPU
            return [combine_first_ax(inp["frms_ev_fast_u8"])]
        keys = {"multi": ("frms_ev_slow_tensor", "frms_ev_fast_tensor"), "single": ("frms_ev_fast_tensor",)}
        if self.comm.path_type not in keys:
            raise NotImplementedError(f"path_type={self.comm.path_type}")
        return [combine_first_ax(inp[k]) for k in keys[self.comm.path_type]]

    def forward_encoder(self, inp):
        maps = self.sf_mdl.forward_features(self.get_feats(inp))
        assert len(maps) == self.sf_mdl.num_pathways
        return maps

    def calibrate_weight_rounding(self, inp):
        """Eval only: measure, on the calibration batch `inp` (the same dict contract as forward; clips other than the
        ones evaluated), the per-channel constants the bf16 rounding of the trunk's convolution weights adds, and fold
        their correction into the BN shifts (`VideoTrunk.calibrate_weight_rounding`).  Free at run time."""
        return self.sf_mdl.calibrate_weight_rounding(self.get_feats(inp))

    def forward_decoder(self, enc_out, inp):
        pooled = self.head(enc_out).permute((0, 2, 3, 4, 1))  # [N, C, 1, 1, 1] -> channels last
        logits = self.proj_head(pooled).view(len(inp["vseg_idx"]), _num_events(inp), -1)
        assert logits.size(-1) == len(self.comm.vb_id_vocab)
        return logits

    def forward(self, inp: Dict):
        return {"mdl_out": self.forward_decoder(self.forward_encoder(inp), inp)}


class _LossBase(nn.Module):
    """`loss_fn = Cls(cfg, comm)`, `.loss_keys`, `loss_fn(out, inp) -> {"loss": 0-dim tensor}` (mdl_sf_base.py:219-243)."""

    loss_keys = ["loss"]

    def __init__(self, cfg, comm):
        super().__init__()
        self.cfg, self.comm = cfg, comm
        self.loss_keys = list(type(self).loss_keys)


class LossB(_LossBase):
    """Mean cross entropy of the per-event verb logits (`:219-231`)."""

    def forward(self, mdl_out, inp):
        flat = combine_first_ax
        return {"loss": hip_cross_entropy(flat(mdl_out["mdl_out"]), flat(inp["label_tensor"]))}


class LossLambda(_LossBase):
    """The model computed its own loss (`:234-243`)."""

    def forward(self, mdl_out, inp):
        if "loss" not in mdl_out:
            raise AssertionError("the model's output carries no 'loss'")
        return {"loss": mdl_out["loss"]}


class TxEncoderNew(TxCodeEnc):
    """mdl_sf_base.py:341-381."""

    def __init__(self, cfg, comm):
        self.full_cfg = cfg
        self.comm = comm
        args = cfg.tx_dec
        super().__init__(d_model=1024, n_vocab_src=0, vocab_trg=0, d_hidden=1024,
                         n_layers=args.encoder_layers, n_heads=args.encoder_attention_heads,
                         drop_ratio=args.dropout, pe=False)

    def forward(self, src_tokens=None, src_lengths=None, return_all_hiddens=False,
                token_embeddings=None):
        assert token_embeddings is not None
        enc_out = self.encoder(token_embeddings)[-1]
        return EncoderOut(encoder_out=enc_out.transpose(0, 1).contiguous(),
                          encoder_padding_mask=None, encoder_embedding=None, encoder_states=None,
                          src_tokens=None, src_lengths=None)


class TxEncoderNew_Conc(TxEncoderOld):
    """mdl_sf_base.py:395-420 (`tx_enc_type: new_conc`): the fairseq-style encoder's output concatenated with
    its input features and mixed by Linear(2d, d)-ReLU-Linear(d, d) (d = 1024 upstream)."""

    def __init__(self, cfg, comm):
        super().__init__(cfg, comm)
        d = cfg.tx_dec.encoder_embed_dim
        self.orig_tx_out_comb = HipMLP(nn.Linear(2 * d, d), nn.ReLU(), nn.Linear(d, d))

    def forward(self, src_tokens=None, src_lengths=None, return_all_hiddens=False, token_embeddings=None):
        tx_out = super().forward(src_tokens=src_tokens, src_lengths=src_lengths,
                                 return_all_hiddens=return_all_hiddens, token_embeddings=token_embeddings)
        enc_out = tx_out.encoder_out.transpose(0, 1)  # B x T x C
        enc_out3 = self.orig_tx_out_comb(torch.cat([token_embeddings.float(), enc_out], dim=-1))
        return EncoderOut(encoder_out=enc_out3.transpose(0, 1).contiguous(), encoder_padding_mask=None,
                          encoder_embedding=None, encoder_states=None, src_tokens=None, src_lengths=None)


def TxEncoder(cfg, comm):
    if cfg.mdl.tx_enc_type == "new":
        return TxEncoderNew(cfg, comm)
    if cfg.mdl.tx_enc_type == "old":  # fairseq TransformerEncoder
        return TxEncoderOld(cfg, comm)
    if cfg.mdl.tx_enc_type == "new_conc":
        return TxEncoderNew_Conc(cfg, comm)
    raise NotImplementedError(f"tx_enc_type={cfg.mdl.tx_enc_type}")


def get_head_dim(full_cfg) -> int:
    d = full_cfg.ds.vsitu.vsit_frm_feats_dir
    if "i3d" in d:
        return 2048
    elif ("slow_fast" in d) or ("sfast" in d):
        return 2304
    raise NotImplementedError


class GPT2_hf_fseqDec(HuggingFaceGPT2Decoder):
    """mdl_sf_base.py:449-455."""

    def __init__(self, cfg, comm):
        self.full_cfg = cfg
        self.comm = comm
        super().__init__(cfg, comm.gpt2_hf_tok)


def TxDecoder(full_cfg, comm):
    """mdl_sf_base.py:458-464."""
    if full_cfg.mdl.tx_dec_type == "gpt2":
        return GPT2_hf_fseqDec(full_cfg, comm)
    if full_cfg.mdl.tx_dec_type == "txdec":  # fairseq TransformerDecoder
        return TxDecoderReal(full_cfg, comm)
    raise NotImplementedError(f"tx_dec_type={full_cfg.mdl.tx_dec_type}")


class Simple_GPT2(nn.Module):
    """mdl_sf_base.py:467-532: the text-only baseline -- GPT-2 fine-tuned as a language model over the SRL
    token sequence of every event (first annotation), greedy generation from its first token."""

    GEN_MAX_LENGTH = 60  # mdl_sf_base.py:497

    def __init__(self, cfg, comm):
        super().__init__()
        self.full_cfg = cfg
        self.cfg = cfg.mdl
        self.comm = comm
        self.build_model()

    def build_model(self):
        self.gpt2_mdl = GPT2LMHeadModelHip(*GPT2_DIMS[self.cfg.gpt2_mdl_name])
        self.voc_size = len(self.comm.gpt2_hf_tok)
        self.gpt2_mdl.resize_token_embeddings(self.voc_size)
        self.pad_index = self.comm.gpt2_hf_tok.pad_token_id
        self.bos_index = self.comm.gpt2_hf_tok.eos_token_id

    def _first_tokens(self, inp):
        src_toks1 = inp["seq_out_by_ev"][:, :, [0], :]
        B, num_ev, num_seq_eg, seq_len = src_toks1.shape
        return src_toks1.reshape(B * num_ev, num_seq_eg * seq_len)[..., :1].contiguous(), (B, num_ev, num_seq_eg)

    def forward_gen(self, inp, *args):
        inp_ids, (B, num_ev, num_seq_eg) = self._first_tokens(inp)
        # huggingface's generate stops a row at the model config's eos id, the tokenizer's eos (= bos here)
        out_sents = self.gpt2_mdl.generate_greedy(inp_ids, self.GEN_MAX_LENGTH, self.pad_index, self.bos_index)
        return out_sents.view(B, num_ev, num_seq_eg, -1)

    def forward(self, inp):
        toks = _first_annotation(inp["seq_out_by_ev"])
        keep = _first_annotation(inp["seq_out_lens_by_ev"]).ne(0)
        if self.training and torch.is_grad_enabled():
            tick = torch.zeros(1, device=toks.device, requires_grad=True)
            logits = _GPT2TrainFn.apply(self.gpt2_mdl, toks, keep, tick)
        else:
            logits = self.gpt2_mdl.forward_logits(toks, keep)
        return {"loss": gpt2_lm_loss(logits, toks, self.pad_index), "logits": logits}


class Simple_GPT2_New(Simple_GPT2):
    """mdl_sf_base.py:560-587 (`new_gpt2_only`): the same model; generation runs to 60 tokens past the prompt.
    (`GPT2_New.prepare_inputs_for_generation`'s `vid_emb` prefix is never passed by this caller.)"""

    GEN_MAX_LENGTH = 60 + 1  # mdl_sf_base.py:579: 60 + inp_ids.size(-1), prompts are one token


class Simple_TxDec(nn.Module):
    """mdl_sf_base.py:595-675: teacher-forced LM loss over the SRL token sequence of every event
    (first annotation only) and beam-search generation from its first token; decoder = GPT-2
    (`tx_dec_type: gpt2`) or the fairseq-style TransformerDecoder (`txdec`), forward and backward on
    the HIP kernels."""

    def __init__(self, cfg, comm):
        super().__init__()
        self.full_cfg, self.cfg, self.sf_cfg, self.comm = cfg, cfg.mdl, cfg.sf_mdl, comm
        self.use_encoder = False  # read by seq_gen.EnsembleModel
        self.build_model()

    def build_model(self):
        tok = self.comm.gpt2_hf_tok
        self.decoder = TxDecoder(self.full_cfg, self.comm)
        self.pad_index, self.bos_index = tok.pad_token_id, tok.eos_token_id
        self.max_decoder_positions = lambda: 1024

    def forward_encoder(self, inp):
        return None  # text only: subclasses supply the video memory

    def prepare_prev_toks_inp(self, inp):
        """Target tokens / lengths / verb tokens of every event's FIRST annotation, events folded into the batch axis."""
        toks = _first_annotation(inp["seq_out_by_ev"])
        lens = _first_annotation(inp["seq_out_lens_by_ev"]).sum(dim=-1)
        return {"dst_toks": toks, "dst_lens": lens, "vb_only_tokens": _first_annotation(inp["vb_out_by_ev"])}

    def forward_decoder(self, prev_tokens, encoder_out, incremental_state=None, temperature=None):
        memory = None if (isinstance(encoder_out, list) and not encoder_out) else encoder_out
        return self.decoder(prev_tokens, encoder_out=memory, incremental_state=incremental_state)

    def forward(self, inp):
        toks = self.prepare_prev_toks_inp(inp)["dst_toks"]
        logits = self.forward_decoder(prev_tokens=toks, encoder_out=self.forward_encoder(inp))[0]
        return {"loss": gpt2_lm_loss(logits, toks, self.pad_index), "logits": logits}

    def forward_gen(self, inp, seq_gen):
        """Beam search from each event's first token; hypotheses padded into [B, E, 1, longest]."""
        prep = self.prepare_prev_toks_inp(inp)
        prompt = prep["dst_toks"][..., :1]
        inp["src_tokens"], inp["src_lengths"] = prompt, prep["dst_lens"]
        hyps = seq_gen._generate(inp, prefix_tokens=prompt)
        best = [h[0]["tokens"] for h in hyps]
        out = prompt.new_full((prompt.size(0), max(len(t) for t in best)), self.pad_index)
        for row, t in enumerate(best):
            out[row, : len(t)] = t
        b, n_ev = inp["seq_out_by_ev"].shape[:2]
        return out.view(b, n_ev, 1, -1)


class Reorderer:
    """mdl_sf_base.py:694-748: beam reorder of an EncoderOut (T x B x C tensors along dim 1)."""

    def reorder_encoder_out(self, encoder_out, new_order):
        sel = lambda t, d: t if t is None else t.index_select(d, new_order)
        states = encoder_out.encoder_states
        if states is not None:
            states = [st.index_select(1, new_order) for st in states]
        return EncoderOut(encoder_out=sel(encoder_out.encoder_out, 1),
                          encoder_padding_mask=sel(encoder_out.encoder_padding_mask, 0),
                          encoder_embedding=sel(encoder_out.encoder_embedding, 0),
                          encoder_states=states, src_tokens=sel(encoder_out.src_tokens, 0),
                          src_lengths=sel(encoder_out.src_lengths, 0))


class SFPreFeats_TxDec(Simple_TxDec, Reorderer):
    """mdl_sf_base.py:763-790 (`sfpret_txed_vbarg`): pre-extracted [B,5,head_dim] features -> vid_feat_encoder
    -> EncoderOut [1, 5B, 1024] (no transformer encoder) -> decoder."""

    def build_model(self):
        super().build_model()
        self.vid_feat_encoder = HipMLP(nn.Linear(get_head_dim(self.full_cfg), 1024), nn.ReLU(),
                                       nn.Linear(1024, 1024))
        self.use_encoder = True

    def forward_encoder(self, inp):
        frm_feats = inp["frm_feats"]
        B, n_ev = inp["vseg_idx"].size(0), frm_feats.size(1)
        out = self.vid_feat_encoder(frm_feats.float()).view(B * n_ev, 1, -1)
        return EncoderOut(encoder_out=out.transpose(0, 1).contiguous(), encoder_padding_mask=None,
                          encoder_embedding=None, encoder_states=None, src_tokens=None, src_lengths=None)


class SFPreFeats_TxEncDec(Simple_TxDec, Reorderer):
    """mdl_sf_base.py:793-832: pre-extracted [B,5,2304] features -> vid_feat_encoder ->
    TxEncoderNew -> EncoderOut [1, 5B, 1024] (each event is its own one-token memory) -> decoder
    (`Simple_TxDec.forward` / `forward_gen`)."""

    def __init__(self, cfg, comm, head_dim=None):
        self._head_dim = head_dim
        super().__init__(cfg, comm)

    def build_model(self):
        super().build_model()
        head_dim = self._head_dim
        if head_dim is None:  # the reference derives it from the feature directory's name
            head_dim = get_head_dim(self.full_cfg)
        self.vid_feat_encoder = HipMLP(nn.Linear(head_dim, 1024), nn.ReLU(), nn.Linear(1024, 1024))
        self.use_encoder = True
        self.vid_feat_txenc = TxEncoder(self.full_cfg, self.comm)

    def forward_encoder(self, inp):
        frm_feats = inp["frm_feats"]
        B = inp["vseg_idx"].size(0)
        n_ev = frm_feats.size(1)
        out = self.vid_feat_encoder(frm_feats.float())
        out = out.view(B, n_ev, -1)
        tx_out = self.vid_feat_txenc(src_tokens=out[..., 0], src_lengths=None,
                                     return_all_hiddens=True, token_embeddings=out)
        enc_out_batch1 = tx_out.encoder_out.transpose(0, 1).contiguous()
        enc_out3 = enc_out_batch1.view(B * n_ev, 1, -1).transpose(0, 1).contiguous()
        return EncoderOut(encoder_out=enc_out3, encoder_padding_mask=None, encoder_embedding=None,
                          encoder_states=None, src_tokens=None, src_lengths=None)

class SFBase_TxEnc(SFBase):
    """BASELINE config 3 ("SlowFast-R50 + 6-layer TxEnc verb-pred"), a composition the
    reference does not hold as one class: SFBase trunk + head ->
    `vid_feat_encoder` (SFPreFeats_TxEncDec, :798-800) -> `TxEncoderNew` over the events of
    each video -> per-event Linear(1024, V).  Every piece is parity-tested on its own."""

    def build_projection_head(self, cfg, out_dim=None):
        if out_dim is None:
            out_dim = len(self.comm.vb_id_vocab)
        din = sum(self.head.dim_in)
        self.vid_feat_encoder = HipMLP(nn.Linear(din, 1024), nn.ReLU(), nn.Linear(1024, 1024))
        self.vid_feat_txenc = TxEncoder(self.full_cfg, self.comm)
        self.proj_head = HipMLP(nn.Linear(1024, out_dim))

    def forward_decoder(self, enc_out, inp):
        head_out = self.head(enc_out)  # [N, C, 1, 1, 1]
        B = len(inp["vseg_idx"])
        n_ev = _num_events(inp)
        feats = head_out.view(B, n_ev, -1)
        tok = self.vid_feat_encoder(feats)
        tx = self.vid_feat_txenc(src_tokens=tok[..., 0], src_lengths=None,
                                 return_all_hiddens=True, token_embeddings=tok)
        ev = tx.encoder_out.transpose(0, 1)  # [B, n_ev, 1024]
        out = self.proj_head(ev.contiguous())
        assert out.size(-1) == len(self.comm.vb_id_vocab)
        return out
