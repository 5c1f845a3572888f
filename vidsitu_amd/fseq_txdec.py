"""fairseq's `TransformerDecoder` as `TxDecoderReal` configures it (`vidsitu_code/mdl_sf_base.py:435-446`,
`configs/vsitu_tx_cfgs/transformer.yaml`) on the HIP kernels: post-norm layers
(self attention -> add & norm -> encoder attention -> add & norm -> relu FFN -> add & norm), embed scale
sqrt(d), sinusoidal positions `padding_idx + 1 + t`, `project_out_dim` d -> decoder_output_dim and an
untied `output_projection`, both without bias.  Parameter names / shapes are fairseq's, so its checkpoints
load (`layers.N.self_attn.q_proj.weight`, ...).

* The three self-attention projections run as ONE GEMM on a concatenated [3d, d] weight (cached copy).
* VidSitu's encoder output has ONE position per decoder row (`SFPreFeats_TxEncDec.forward_encoder`,
  `mdl_sf_base.py:806-832`: `[1, B*n_ev, 1024]`), so the softmax over encoder positions is exactly 1 and
  encoder attention is `out_proj(v_proj(enc))` broadcast over the target positions -- a per-row constant
  (its q / k projections get exactly zero gradient, as autograd gives them).  Longer encoder outputs raise.
* Training: one autograd node -- forward keeps the activations, a hand-written backward writes every
  parameter gradient (overwrite semantics) and returns the gradient of the encoder output.  Dropout
  (`dropout` after the embedding and after each sub-layer; attention / activation dropout are 0 in the
  reference's YAML) through mask operands of the fused add + layernorm kernel.
* Generation: a KV cache per layer read through the beam ancestry table (`vs_attn_decode`), the encoder
  attention constant computed once per generation (`begin_incremental`), so decode steps never read the
  encoder output and can be captured in hipGraphs (`seq_gen._DeviceSearchSession`).
"""
import math

import torch
from torch import nn

from . import ops
from .hf_gpt2_fseq import KVCacheState
from .transformer_code import AddLayerNormFn, AttnSmallFn, hip_linear


def _xavier(out_f, in_f, gain=1.0):
    t = torch.empty(out_f, in_f)
    nn.init.xavier_uniform_(t, gain=gain)
    return t


class TransformerDecoderHip(nn.Module):
    def __init__(self, vocab, d_model, ffn, n_head, n_layer, out_dim, pad, dropout=0.1, max_positions=1024):
        super().__init__()
        self.vocab, self.d_model, self.ffn, self.n_head, self.n_layer = vocab, d_model, ffn, n_head, n_layer
        self.out_dim, self.pad, self.p_drop, self.max_positions = out_dim, pad, float(dropout), max_positions
        p = {}
        emb = torch.randn(vocab, d_model) * d_model ** -0.5
        emb[pad] = 0
        p["embed_tokens.weight"] = emb
        for i in range(n_layer):
            q = f"layers.{i}."
            for att in ("self_attn", "encoder_attn"):
                for pr in ("k_proj", "v_proj", "q_proj", "out_proj"):  # MultiheadAttention.reset_parameters
                    p[q + f"{att}.{pr}.weight"] = _xavier(d_model, d_model,
                                                          1.0 if pr == "out_proj" else 1 / math.sqrt(2))
                    p[q + f"{att}.{pr}.bias"] = torch.zeros(d_model)
                p[q + f"{att}_layer_norm.weight"] = torch.ones(d_model)
                p[q + f"{att}_layer_norm.bias"] = torch.zeros(d_model)
            p[q + "fc1.weight"], p[q + "fc1.bias"] = _xavier(ffn, d_model), torch.zeros(ffn)
            p[q + "fc2.weight"], p[q + "fc2.bias"] = _xavier(d_model, ffn), torch.zeros(d_model)
            p[q + "final_layer_norm.weight"] = torch.ones(d_model)
            p[q + "final_layer_norm.bias"] = torch.zeros(d_model)
        p["project_out_dim.weight"] = _xavier(out_dim, d_model)
        p["output_projection.weight"] = torch.randn(vocab, out_dim) * out_dim ** -0.5
        self._names = list(p)
        for k, v in p.items():
            self.register_parameter(k.replace(".", "__"), nn.Parameter(v))
        self._cat, self._tab = {}, {}
        self._wt_epoch = 0  # bumped when cached weight copies are dropped (captured decode graphs go stale)
        self._saved = None

    # ---- fairseq-compatible state dict -------------------------------------------------------------
    def P(self, name):
        return getattr(self, name.replace(".", "__"))

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        for k in self._names:
            v = self.P(k)
            destination[prefix + k] = v if keep_vars else v.detach()
        dev = self.P("embed_tokens.weight").device
        destination[prefix + "embed_positions._float_tensor"] = torch.zeros(1, device=dev)
        destination[prefix + "version"] = torch.tensor([3.0], device=dev)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        mine = {k[len(prefix):]: v for k, v in state_dict.items() if k.startswith(prefix)}
        with torch.no_grad():
            for k in self._names:
                if k not in mine:
                    missing_keys.append(prefix + k)
                elif tuple(mine[k].shape) != tuple(self.P(k).shape):
                    error_msgs.append(f"size mismatch for {prefix + k}: {tuple(mine[k].shape)} vs "
                                      f"{tuple(self.P(k).shape)}")
                else:
                    self.P(k).copy_(torch.as_tensor(mine[k]))
        for k in mine:
            if k not in self._names and k not in ("embed_positions._float_tensor", "version"):
                unexpected_keys.append(prefix + k)
        self._drop()

    def _drop(self):
        self._cat.clear()
        self._wt_epoch += 1

    def _guard_capture(self):
        if torch.cuda.is_current_stream_capturing():
            raise ops._lib.VsError("decoder weight copies must exist before a graph capture (run one eager step)")

    def _qkv(self, i):
        """Concatenated self-attention projection [3d, d] / [3d] of layer i (q | k | v)."""
        t = self._cat.get(i)
        w0 = self.P(f"layers.{i}.self_attn.q_proj.weight")
        if t is None or t[0].device != w0.device:
            self._guard_capture()
            q = f"layers.{i}.self_attn."
            t = (torch.cat([self.P(q + f"{n}_proj.weight").detach() for n in "qkv"]).contiguous(),
                 torch.cat([self.P(q + f"{n}_proj.bias").detach() for n in "qkv"]).contiguous())
            self._cat[i] = t
        return t

    def _pos_table(self, device):
        """Row 0 = zeros (padding); row 1 + t = the sinusoid of position padding_idx + 1 + t, computed on
        the CPU in fp32 in the order of fairseq's `SinusoidalPositionalEmbedding.get_embedding`."""
        t = self._tab.get(str(device))
        if t is None:
            self._guard_capture()
            d, half = self.d_model, self.d_model // 2
            freq = torch.exp(torch.arange(half, dtype=torch.float) * -(math.log(10000) / (half - 1)))
            pos = torch.arange(self.pad + 1, self.pad + 1 + self.max_positions, dtype=torch.float)
            ang = pos.unsqueeze(1) * freq.unsqueeze(0)
            tab = torch.cat([torch.sin(ang), torch.cos(ang)], dim=1)
            if d % 2 == 1:
                tab = torch.cat([tab, torch.zeros(tab.shape[0], 1)], dim=1)
            t = torch.cat([torch.zeros(1, d), tab]).contiguous().to(device)
            self._tab[str(device)] = t
        return t

    def _embed(self, tokens, pos0=None):
        """tokens i64 [R, L] -> f32 [R*L, d]; pos0: every token sits at target position pos0 (one
        incremental step, fairseq's `positions[:, -1:]`), else `utils.make_positions`."""
        if pos0 is None:
            mask = tokens.ne(self.pad)
            idx = torch.cumsum(mask, dim=1) * mask
        else:
            idx = torch.full_like(tokens, pos0 + 1)
        return ops.embed_pos_fwd(tokens, self.P("embed_tokens.weight"), self._pos_table(tokens.device), idx,
                                 math.sqrt(self.d_model))

    def _cross_const(self, i, enc2d):
        """Encoder attention over ONE encoder position: out_proj(v_proj(enc)) per row -> ([R, d], cv)."""
        q = f"layers.{i}.encoder_attn."
        cv = ops.gemm_nt(enc2d, self.P(q + "v_proj.weight"), self.P(q + "v_proj.bias"))
        return ops.gemm_nt(cv, self.P(q + "out_proj.weight"), self.P(q + "out_proj.bias")), cv

    @staticmethod
    def enc_rows(encoder_out, rows):
        """EncoderOut / tensor [S, R, d] -> [R, d] (S must be 1), or None."""
        if encoder_out is None:
            return None
        enc = encoder_out.encoder_out if hasattr(encoder_out, "encoder_out") else encoder_out
        if enc.dim() != 3 or enc.shape[0] != 1 or enc.shape[1] != rows:
            raise NotImplementedError(f"encoder attention over {tuple(enc.shape)}: only one encoder position "
                                      "per decoder row is built (VidSitu's SFPreFeats_TxEncDec)")
        return enc[0].float().contiguous()

    # ---- whole-sequence pass ------------------------------------------------------------------------
    def _forward_seq(self, tokens, enc2d, train):
        if not tokens.is_cuda:
            raise ops._lib.VsError("the decoder runs on the HIP kernels only (GPU tensor required)")
        r, l = tokens.shape
        d, p = self.d_model, self.p_drop if train else 0.0
        km = tokens.ne(self.pad).to(torch.uint8).contiguous()

        def mask():
            return ops.dropout_mask((r * l, d), p, tokens.device) if p > 0 else None

        x = self._embed(tokens)
        m0 = mask()
        if m0 is not None:
            x = x * m0
        layers = []
        for i in range(self.n_layer):
            q = f"layers.{i}."
            wqkv, bqkv = self._qkv(i)
            qkv = ops.gemm_nt(x, wqkv, bqkv)
            o = ops.attn_causal(qkv, km, r, l, self.n_head)
            sa = ops.gemm_nt(o, self.P(q + "self_attn.out_proj.weight"), self.P(q + "self_attn.out_proj.bias"))
            m1 = mask()
            x1, mu1, rs1 = ops.add_layernorm_fwd(x, sa, self.P(q + "self_attn_layer_norm.weight"),
                                                 self.P(q + "self_attn_layer_norm.bias"), 1e-5, rmask=m1)
            if enc2d is not None:
                co, cv = self._cross_const(i, enc2d)
                ca = co.repeat_interleave(l, dim=0)
                m2 = mask()
                x2, mu2, rs2 = ops.add_layernorm_fwd(x1, ca, self.P(q + "encoder_attn_layer_norm.weight"),
                                                     self.P(q + "encoder_attn_layer_norm.bias"), 1e-5, rmask=m2)
            else:
                cv = ca = m2 = mu2 = rs2 = None
                x2 = x1
            f1 = ops.gemm_nt(x2, self.P(q + "fc1.weight"), self.P(q + "fc1.bias"), act=ops.ACT_RELU)
            f2 = ops.gemm_nt(f1, self.P(q + "fc2.weight"), self.P(q + "fc2.bias"))
            m3 = mask()
            x3, mu3, rs3 = ops.add_layernorm_fwd(x2, f2, self.P(q + "final_layer_norm.weight"),
                                                 self.P(q + "final_layer_norm.bias"), 1e-5, rmask=m3)
            if train:
                layers.append((x, qkv, o, sa, m1, mu1, rs1, x1, cv, ca, m2, mu2, rs2, x2, f1, f2, m3, mu3, rs3))
            x = x3
        po = ops.gemm_nt(x, self.P("project_out_dim.weight"))
        logits = ops.gemm_nt(po, self.P("output_projection.weight"))
        if train:
            self._saved = (tokens, km, r, l, enc2d, m0, layers, x, po)
        return logits.view(r, l, -1)

    @torch.no_grad()
    def forward_logits(self, tokens, enc2d=None):
        """tokens i64 [R, L], enc2d f32 [R, d] or None -> logits f32 [R, L, V] (no dropout)."""
        return self._forward_seq(tokens, enc2d, train=False)

    def forward_train(self, tokens, enc2d=None):
        return self._forward_seq(tokens, enc2d, train=True)

    # ---- backward ------------------------------------------------------------------------------------
    def _grad(self, name):
        p = self.P(name)
        if p.grad is None:
            p.grad = torch.empty_like(p)
        return p.grad

    def _lin_bwd(self, name, x, dy, bias=True, need_dx=True):
        """y = x @ W^T + b with fairseq's W [out, in]: dW = dy^T x, db = colsum(dy), dx = dy @ W."""
        ops.gemm_nt(ops.transpose_f32(dy), ops.transpose_f32(x), out=self._grad(name + ".weight"))
        if bias:
            ops.colsum_f32(dy, out=self._grad(name + ".bias"))
        return ops.gemm_nt(dy, ops.transpose_f32(self.P(name + ".weight"))) if need_dx else None

    @torch.no_grad()
    def backward(self, dlogits):
        """dlogits [R*L, V] -> every parameter's .grad (overwritten); returns d(enc2d) [R, d] or None."""
        tokens, km, r, l, enc2d, m0, layers, x_last, po = self._saved
        self._saved = None
        d = self.d_model
        dlogits = dlogits.reshape(r * l, -1)
        dpo = self._lin_bwd("output_projection", po, dlogits, bias=False)
        dx = self._lin_bwd("project_out_dim", x_last, dpo, bias=False)
        denc = None
        for i in reversed(range(self.n_layer)):
            q = f"layers.{i}."
            x, qkv, o, sa, m1, mu1, rs1, x1, cv, ca, m2, mu2, rs2, x2, f1, f2, m3, mu3, rs3 = layers[i]
            dx2_a, df2, _, _ = ops.add_layernorm_bwd(dx, x2, f2, self.P(q + "final_layer_norm.weight"), mu3, rs3,
                                                     rmask=m3, dg_out=self._grad(q + "final_layer_norm.weight"),
                                                     db_out=self._grad(q + "final_layer_norm.bias"))
            df1 = ops.relu_bwd(self._lin_bwd(q + "fc2", f1, df2), f1)
            dx2 = ops.add_f32(dx2_a, self._lin_bwd(q + "fc1", x2, df1))
            if enc2d is not None:
                dx1, dca, _, _ = ops.add_layernorm_bwd(dx2, x1, ca, self.P(q + "encoder_attn_layer_norm.weight"),
                                                       mu2, rs2, rmask=m2,
                                                       dg_out=self._grad(q + "encoder_attn_layer_norm.weight"),
                                                       db_out=self._grad(q + "encoder_attn_layer_norm.bias"))
                dco = dca.view(r, l, d).sum(dim=1)
                dcv = self._lin_bwd(q + "encoder_attn.out_proj", cv, dco)
                de = self._lin_bwd(q + "encoder_attn.v_proj", enc2d, dcv)
                denc = de if denc is None else ops.add_f32(denc, de)
                for pr in ("q_proj", "k_proj"):  # softmax over one key: no gradient reaches q / k
                    self._grad(q + f"encoder_attn.{pr}.weight").zero_()
                    self._grad(q + f"encoder_attn.{pr}.bias").zero_()
            else:
                dx1 = dx2
                for n in self._names:  # unused sub-layer: zero gradients (overwrite semantics)
                    if n.startswith(q + "encoder_attn"):
                        self._grad(n).zero_()
            dx_a, dsa, _, _ = ops.add_layernorm_bwd(dx1, x, sa, self.P(q + "self_attn_layer_norm.weight"), mu1, rs1,
                                                    rmask=m1, dg_out=self._grad(q + "self_attn_layer_norm.weight"),
                                                    db_out=self._grad(q + "self_attn_layer_norm.bias"))
            do = self._lin_bwd(q + "self_attn.out_proj", o, dsa)
            dqkv = ops.attn_causal_bwd(qkv, km, do, r, l, self.n_head)
            wqkv, _ = self._qkv(i)
            dw = ops.gemm_nt(ops.transpose_f32(dqkv), ops.transpose_f32(x))  # [3d, d]
            db = ops.colsum_f32(dqkv)
            for j, n in enumerate("qkv"):
                self._grad(q + f"self_attn.{n}_proj.weight").copy_(dw[j * d:(j + 1) * d])
                self._grad(q + f"self_attn.{n}_proj.bias").copy_(db[j * d:(j + 1) * d])
            dx = ops.add_f32(dx_a, ops.gemm_nt(dqkv, ops.transpose_f32(wqkv)))
        if m0 is not None:
            dx = dx * m0
        demb = self._grad("embed_tokens.weight")
        demb.zero_()
        ops.embed_scatter_bwd(tokens, dx, demb, math.sqrt(d), self.pad)
        self._drop()  # the optimizer is about to change the parameters
        return denc

    # ---- incremental decoding --------------------------------------------------------------------------
    def begin_incremental(self, state, enc2d, rows, max_len):
        """(Re)initialise an incremental state for `rows` hypotheses: caches allocated once, the
        encoder-attention constants of this generation written into static buffers."""
        dev = self.P("embed_tokens.weight").device
        dh = self.d_model // self.n_head
        if state.k is None or state.k.shape[1] != rows or state.k.shape[3] != max_len:
            shape = (self.n_layer, rows, self.n_head, max_len, dh)
            state.k = torch.zeros(shape, dtype=torch.float32, device=dev)
            state.v = torch.zeros(shape, dtype=torch.float32, device=dev)
            state.k2 = state.v2 = None
            state.cross = None
        state.len = 0
        if enc2d is None:
            state.cross = None
        else:
            if getattr(state, "cross", None) is None:
                state.cross = torch.empty((self.n_layer, rows, self.d_model), dtype=torch.float32, device=dev)
            for i in range(self.n_layer):
                state.cross[i].copy_(self._cross_const(i, enc2d)[0])
        state.ready = True

    @torch.no_grad()
    def forward_step(self, last_tokens, state):
        """One cached step: last_tokens i64 [rows] at target position state.len -> logits [rows, V]."""
        rows = last_tokens.shape[0]
        t = state.len
        x = self._embed(last_tokens.view(rows, 1), pos0=t)
        for i in range(self.n_layer):
            q = f"layers.{i}."
            wqkv, bqkv = self._qkv(i)
            qkv = ops.gemm_nt(x, wqkv, bqkv)
            o = ops.attn_decode(qkv, state.k[i][:rows], state.v[i][:rows], None, t, ancestry=state.anc)
            sa = ops.gemm_nt(o, self.P(q + "self_attn.out_proj.weight"), self.P(q + "self_attn.out_proj.bias"))
            x = ops.add_layernorm_fwd(x, sa, self.P(q + "self_attn_layer_norm.weight"),
                                      self.P(q + "self_attn_layer_norm.bias"), 1e-5)[0]
            if state.cross is not None:
                x = ops.add_layernorm_fwd(x, state.cross[i][:rows], self.P(q + "encoder_attn_layer_norm.weight"),
                                          self.P(q + "encoder_attn_layer_norm.bias"), 1e-5)[0]
            f1 = ops.gemm_nt(x, self.P(q + "fc1.weight"), self.P(q + "fc1.bias"), act=ops.ACT_RELU)
            f2 = ops.gemm_nt(f1, self.P(q + "fc2.weight"), self.P(q + "fc2.bias"))
            x = ops.add_layernorm_fwd(x, f2, self.P(q + "final_layer_norm.weight"),
                                      self.P(q + "final_layer_norm.bias"), 1e-5)[0]
        state.len = t + 1
        return ops.gemm_nt(ops.gemm_nt(x, self.P("project_out_dim.weight")), self.P("output_projection.weight"))

    def reorder_state(self, state, new_order):
        """fairseq reorder_incremental_state (physical gather; the device-side search uses the ancestry
        table instead): row r becomes old row new_order[r]."""
        if state.k is None:
            return
        if state.anc is not None:
            raise ops._lib.VsError("this cache is reordered through its ancestry table (vs_beam_step)")
        rows = new_order.numel()
        if state.k2 is None:
            state.k2, state.v2 = torch.empty_like(state.k), torch.empty_like(state.v)
        for li in range(self.n_layer):
            ops.kv_gather(state.k[li], state.k2[li], new_order, state.len)
            ops.kv_gather(state.v[li], state.v2[li], new_order, state.len)
        state.k, state.k2 = state.k2, state.k
        state.v, state.v2 = state.v2, state.v
        if getattr(state, "cross", None) is not None:
            state.cross = state.cross.index_select(1, new_order)
        state.rows = rows


class _TxDecTrainFn(torch.autograd.Function):
    """logits = decoder(tokens, enc) as one autograd node (`enc` may be None)."""

    @staticmethod
    def forward(ctx, model, tokens, enc2d, _tick):
        ctx.model = model
        ctx.has_enc = enc2d is not None
        out = model.forward_train(tokens, None if enc2d is None else enc2d.detach())
        # the saved activations of THIS forward travel with the node (several forwards may precede a backward)
        ctx.saved_state, model._saved = model._saved, None
        return out

    @staticmethod
    def backward(ctx, dlogits):
        ctx.model._saved = ctx.saved_state
        denc = ctx.model.backward(dlogits.contiguous())
        return None, None, (denc if ctx.has_enc else None), None


class TxDecoderReal(nn.Module):
    """`TxDecoderReal(TransformerDecoder)` (`mdl_sf_base.py:435-446`) with fairseq's decoder call surface:
    `forward(prev_output_tokens, encoder_out=, incremental_state=) -> (logits,)`,
    `reorder_incremental_state`, `max_positions`."""

    uses_encoder_out = False  # decode steps read the constants `begin_incremental` stored in the state

    def __init__(self, cfg, comm):
        super().__init__()
        self.full_cfg, self.comm = cfg, comm
        tok = comm.gpt2_hf_tok
        a = cfg.tx_dec
        self.pad_idx = tok.pad_token_id
        self.model = TransformerDecoderHip(len(tok), a.decoder_embed_dim, a.decoder_ffn_embed_dim,
                                           a.decoder_attention_heads, a.decoder_layers, a.decoder_output_dim,
                                           self.pad_idx, dropout=a.dropout)

    def begin_incremental(self, state, encoder_out, rows, max_len):
        self.model.begin_incremental(state, self.model.enc_rows(encoder_out, rows), rows, max_len)

    def forward(self, prev_output_tokens, src_lengths=None, incremental_state=None, encoder_out=None):
        m = self.model
        if incremental_state:  # KVCacheState is truthy; fairseq's dict-based incremental decoding
            st = incremental_state
            if not getattr(st, "ready", False):
                self.begin_incremental(st, encoder_out, prev_output_tokens.size(0),
                                       getattr(st, "max_len", None) or m.max_positions)
            if st.len != prev_output_tokens.size(1) - 1:
                raise ops._lib.VsError("incremental state out of step with prev_output_tokens")
            return (m.forward_step(prev_output_tokens[:, -1].contiguous(), st).unsqueeze(1),)
        enc2d = m.enc_rows(encoder_out, prev_output_tokens.size(0))
        if self.training and torch.is_grad_enabled():
            tick = torch.zeros(1, device=prev_output_tokens.device, requires_grad=True)
            return (_TxDecTrainFn.apply(m, prev_output_tokens, enc2d, tick),)
        return (m.forward_logits(prev_output_tokens, enc2d),)

    def reorder_incremental_state(self, incremental_state, new_order):
        if incremental_state:
            self.model.reorder_state(incremental_state, new_order)

    def max_positions(self):
        return self.model.max_positions

    def max_decoder_positions(self):
        return self.model.max_positions


# ================================================================================================
# fairseq TransformerEncoder as `TxEncoderOld` uses it (`vidsitu_code/mdl_sf_base.py:246-338`,
# `tx_enc_type: old`): the per-event video features ARE the token embeddings; `src_tokens` (their first
# channel, a float tensor) only drives the position / padding rules.  Post-norm layers, no final norm.
# Built from the autograd functions of the TxEncoderNew path (fused linear, L <= 16 attention, fused
# residual + dropout + layernorm), so the backward is theirs.
# ================================================================================================
def sinusoid_table(pad, n_rows, d, device):
    """Row 0 = zeros; row 1 + t = fairseq's sinusoidal embedding of position pad + 1 + t (CPU fp32 math)."""
    half = d // 2
    freq = torch.exp(torch.arange(half, dtype=torch.float) * -(math.log(10000) / (half - 1)))
    ang = torch.arange(pad + 1, pad + 1 + n_rows, dtype=torch.float).unsqueeze(1) * freq.unsqueeze(0)
    tab = torch.cat([torch.sin(ang), torch.cos(ang)], dim=1)
    if d % 2 == 1:
        tab = torch.cat([tab, torch.zeros(n_rows, 1)], dim=1)
    return torch.cat([torch.zeros(1, d), tab]).contiguous().to(device)


class _FseqSelfAttn(nn.Module):
    def __init__(self, d, heads):
        super().__init__()
        self.k_proj, self.v_proj = nn.Linear(d, d), nn.Linear(d, d)
        self.q_proj, self.out_proj = nn.Linear(d, d), nn.Linear(d, d)
        for m in (self.k_proj, self.v_proj, self.q_proj):
            nn.init.xavier_uniform_(m.weight, gain=1 / math.sqrt(2))
        nn.init.xavier_uniform_(self.out_proj.weight)
        nn.init.zeros_(self.out_proj.bias)
        self.heads, self.scale = heads, math.sqrt(d // heads)

    def forward(self, x):
        o = AttnSmallFn.apply(hip_linear(self.q_proj, x), hip_linear(self.k_proj, x), hip_linear(self.v_proj, x),
                              self.heads, self.scale, None)
        return hip_linear(self.out_proj, o)


class _FseqEncoderLayer(nn.Module):
    def __init__(self, d, ffn, heads, dropout):
        super().__init__()
        self.self_attn = _FseqSelfAttn(d, heads)
        self.self_attn_layer_norm = nn.LayerNorm(d)
        self.fc1, self.fc2 = nn.Linear(d, ffn), nn.Linear(ffn, d)
        self.final_layer_norm = nn.LayerNorm(d)
        self.p = float(dropout)

    def _mask(self, x):
        if self.training and self.p > 0:
            return ops.dropout_mask((x.numel() // x.shape[-1], x.shape[-1]), self.p, x.device)
        return None

    def forward(self, x):
        a = self.self_attn(x)
        ln = self.self_attn_layer_norm
        x = AddLayerNormFn.apply(x, a, ln.weight, ln.bias, ln.eps, self._mask(a), None)
        f = hip_linear(self.fc2, hip_linear(self.fc1, x, relu=True))
        ln = self.final_layer_norm
        return AddLayerNormFn.apply(x, f, ln.weight, ln.bias, ln.eps, self._mask(f), None)


class TxEncoderOld(nn.Module):
    def __init__(self, cfg, comm):
        super().__init__()
        from .mdl_sf_base import EncoderOut

        self._EncoderOut = EncoderOut
        self.full_cfg, self.comm = cfg, comm
        tok = comm.gpt2_hf_tok  # comm[comm.dct_id]; the vb_arg task's dictionary (dat_loader.py:61)
        a = cfg.tx_dec
        d = a.encoder_embed_dim
        self.padding_idx = tok.pad_token_id
        # constructed by the reference and part of its checkpoints, never used (token_embeddings are given)
        self.embed_tokens = nn.Embedding(len(tok), d, self.padding_idx)
        self.embed_tokens.weight.requires_grad_(False)
        self.embed_scale = math.sqrt(d)
        self.p = float(a.dropout)
        self.layers = nn.ModuleList([_FseqEncoderLayer(d, a.encoder_ffn_embed_dim, a.encoder_attention_heads,
                                                       a.dropout) for _ in range(a.encoder_layers)])
        self._tab = {}

    def forward(self, src_tokens=None, src_lengths=None, return_all_hiddens=False, token_embeddings=None):
        assert token_embeddings is not None, "TxEncoderOld is called with the video features as embeddings"
        x = token_embeddings.float()
        if not x.is_cuda:
            raise ops._lib.VsError("the encoder runs on the HIP kernels only (GPU tensor required)")
        b, l, d = x.shape
        tab = self._tab.get(str(x.device))
        if tab is None or tab.shape[0] < l + 1:
            tab = self._tab[str(x.device)] = sinusoid_table(self.padding_idx, max(l, 16), d, x.device)
        mask = src_tokens.ne(self.padding_idx)
        idx = torch.cumsum(mask, dim=1) * mask  # utils.make_positions, as a row of the table
        embed = self.embed_scale * x
        x = embed + tab[idx]
        if self.training and self.p > 0:
            x = x * ops.dropout_mask((b, l, d), self.p, x.device)
        states = [] if return_all_hiddens else None
        for layer in self.layers:
            x = layer(x)
            if states is not None:
                states.append(x.transpose(0, 1))
        return self._EncoderOut(encoder_out=x.transpose(0, 1).contiguous(),
                                encoder_padding_mask=src_tokens.eq(self.padding_idx), encoder_embedding=embed,
                                encoder_states=states, src_tokens=None, src_lengths=None)
