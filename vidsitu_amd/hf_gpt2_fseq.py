"""Host-side mirror of the reference's GPT-2 decoder wrapper, `HuggingFaceGPT2Decoder`
(`vidsitu_code/hf_gpt2_fseq.py:124-215`), on the HIP kernels of csrc/gpt2_ops.hip.

The reference builds `GPT2LMHeadModel.from_pretrained(cfg.mdl.gpt2_mdl_name)` (huggingface
transformers==3.3.1) and resizes its embeddings to `len(comm.gpt2_hf_tok)`; there is no network
here, so `GPT2LMHeadModelHip` is built from the architecture's dimensions with random weights and
loads a huggingface state dict when one is given (same keys and shapes: `transformer.wte.weight`,
`transformer.h.{i}.attn.c_attn.weight` [in, out] Conv1D layout, ..., `lm_head.weight` tied).
Forward semantics kept: `attention_mask = tokens != pad` (`:185`), default position ids (`:187-193`
stays commented out in the reference), `encoder_out` is accepted and unused exactly as stock GPT-2
(no cross-attention) ignores `encoder_hidden_states` (`:198-204`), returns `(lm_logits,)`.

Incremental decoding: the reference passes empty (falsy) `incremental_state` dicts, so it
re-encodes the whole prefix every step.  The same call with
`incremental_state=None` does that here; given a `KVCacheState` the decoder appends one position to
a KV cache instead -- identical arithmetic, one token of work per step.
"""
import math

import torch
from torch import nn

from . import ops

GPT2_DIMS = {
    # name: (n_layer, d_model, n_head, n_positions, vocab)
    "gpt2": (12, 768, 12, 1024, 50257),
    "gpt2-medium": (24, 1024, 16, 1024, 50257),
    "gpt2-synth-tiny": (2, 64, 4, 64, 97),  # tests / smoke only
}


class KVCacheState:
    """Truthy incremental state: per-layer K/V caches [rows, H, Lmax, dh] (double-buffered for the
    beam reorder) and the number of cached positions."""

    def __init__(self):
        self.k = self.v = self.k2 = self.v2 = None
        self.len = 0
        # i32 [rows, Lmax] or None: cache row holding position j of row r (device-side beam search:
        # vs_beam_step permutes this table instead of gathering the cache)
        self.anc = None

    def __bool__(self):
        return True


class GPT2LMHeadModelHip(nn.Module):
    def __init__(self, n_layer, d_model, n_head, n_positions, vocab_size):
        super().__init__()
        self.n_layer, self.d_model, self.n_head, self.n_positions = n_layer, d_model, n_head, n_positions
        self.config = type("Cfg", (), {"n_positions": n_positions, "n_embd": d_model,
                                       "n_layer": n_layer, "n_head": n_head,
                                       "vocab_size": vocab_size})()
        p = {}

        def add(name, *shape, std=0.02, ones=False):
            t = torch.ones(*shape) if ones else (torch.randn(*shape) * std if std else torch.zeros(*shape))
            p[name] = nn.Parameter(t)

        add("transformer.wte.weight", vocab_size, d_model)
        add("transformer.wpe.weight", n_positions, d_model, std=0.01)
        for i in range(n_layer):
            q = f"transformer.h.{i}."
            for ln in ("ln_1", "ln_2"):
                add(q + ln + ".weight", d_model, ones=True)
                add(q + ln + ".bias", d_model, std=0)
            add(q + "attn.c_attn.weight", d_model, 3 * d_model)
            add(q + "attn.c_attn.bias", 3 * d_model, std=0)
            add(q + "attn.c_proj.weight", d_model, d_model)
            add(q + "attn.c_proj.bias", d_model, std=0)
            add(q + "mlp.c_fc.weight", d_model, 4 * d_model)
            add(q + "mlp.c_fc.bias", 4 * d_model, std=0)
            add(q + "mlp.c_proj.weight", 4 * d_model, d_model)
            add(q + "mlp.c_proj.bias", d_model, std=0)
        add("transformer.ln_f.weight", d_model, ones=True)
        add("transformer.ln_f.bias", d_model, std=0)
        # flat registration under the huggingface names ('.' is not allowed in parameter names)
        self._names = list(p)
        for k, v in p.items():
            self.register_parameter(k.replace(".", "__"), v)
        self._wt = {}  # K-contiguous ([out][in]) copies of the Conv1D weights for the kernels
        self._wt_epoch = 0  # bumped whenever the copies are dropped (captured decode graphs go stale)
        self._wpk = {}  # fragment-major copies for the decode-step GEMMs (ops.gemm_nt_packed)

    # ---- huggingface-compatible state dict -------------------------------------------------
    def P(self, name):
        return getattr(self, name.replace(".", "__"))

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        for k in self._names:
            v = self.P(k)
            destination[prefix + k] = v if keep_vars else v.detach()
        w = self.P("transformer.wte.weight")
        destination[prefix + "lm_head.weight"] = w if keep_vars else w.detach()  # tied

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys,
                              unexpected_keys, error_msgs):
        mine = {k[len(prefix):]: v for k, v in state_dict.items() if k.startswith(prefix)}
        with torch.no_grad():
            for k in self._names:
                if k in mine:
                    src = torch.as_tensor(mine[k])
                    if src.shape != self.P(k).shape:
                        error_msgs.append(f"size mismatch for {prefix + k}: {tuple(src.shape)} vs "
                                          f"{tuple(self.P(k).shape)}")
                    else:
                        self.P(k).copy_(src)
                else:
                    missing_keys.append(prefix + k)
        for k in mine:
            # huggingface checkpoints also carry the causal-mask buffers of every block
            if k not in self._names and k != "lm_head.weight" and not k.endswith(".attn.bias") \
                    and not k.endswith(".attn.masked_bias"):
                unexpected_keys.append(prefix + k)
        self._drop_wt()

    def resize_token_embeddings(self, new_size):
        """transformers PreTrainedModel.resize_token_embeddings: keep the first rows, initialise
        the added ones N(0, 0.02) (the reference adds the SRL / separator tokens, :151-152)."""
        old = self.P("transformer.wte.weight")
        if new_size == old.shape[0]:
            return
        new = torch.randn(new_size, old.shape[1], device=old.device) * 0.02
        n = min(new_size, old.shape[0])
        new[:n] = old.detach()[:n]
        setattr(self, "transformer__wte__weight", nn.Parameter(new))
        self.config.vocab_size = new_size

    def _drop_wt(self):
        self._wt.clear()
        self._wpk.clear()
        self._wt_epoch += 1

    def _wp(self, name, transposed=True):
        """Fragment-major copy of a weight ([out][in] orientation) for the 17..64-row decode GEMMs."""
        t = self._wpk.get(name)
        w = self.P(name)
        if t is None or t.device != w.device:
            if torch.cuda.is_current_stream_capturing():
                raise ops._lib.VsError("GPT-2 weight copies must exist before a graph capture "
                                       "(run one eager step first)")
            t = ops.pack_rows_f32(self._w(name) if transposed else w.detach())
            self._wpk[name] = t
        return t

    def _w(self, name):
        """[out][in] view of a Conv1D weight (transposed once, after load / device move)."""
        w = self.P(name)
        t = self._wt.get(name)
        if t is None or t.device != w.device:
            if torch.cuda.is_current_stream_capturing():
                raise ops._lib.VsError("GPT-2 weight copies must exist before a graph capture "
                                       "(run one eager step first)")
            t = w.detach().t().contiguous()
            self._wt[name] = t
        return t

    # ---- compute ---------------------------------------------------------------------------
    def _block(self, i, h, attn_fn):
        q = f"transformer.h.{i}."
        a = ops.add_layernorm_fwd(h, None, self.P(q + "ln_1.weight"), self.P(q + "ln_1.bias"), 1e-5)[0]
        qkv = ops.gemm_nt(a, self._w(q + "attn.c_attn.weight"), self.P(q + "attn.c_attn.bias"))
        o = attn_fn(i, qkv)
        h = ops.gemm_nt(o, self._w(q + "attn.c_proj.weight"), self.P(q + "attn.c_proj.bias"), res=h)
        m = ops.add_layernorm_fwd(h, None, self.P(q + "ln_2.weight"), self.P(q + "ln_2.bias"), 1e-5)[0]
        f = ops.gemm_nt(m, self._w(q + "mlp.c_fc.weight"), self.P(q + "mlp.c_fc.bias"),
                        act=ops.ACT_GELU_NEW)
        return ops.gemm_nt(f, self._w(q + "mlp.c_proj.weight"), self.P(q + "mlp.c_proj.bias"), res=h)

    def _head(self, h):
        hf = ops.add_layernorm_fwd(h, None, self.P("transformer.ln_f.weight"),
                                   self.P("transformer.ln_f.bias"), 1e-5)[0]
        return ops.gemm_nt(hf, self.P("transformer.wte.weight"))

    @torch.no_grad()
    def forward_logits(self, tokens, attention_mask=None):
        """tokens i64 [R, L] -> logits f32 [R, L, V] (whole-sequence pass)."""
        if not tokens.is_cuda:
            raise ops._lib.VsError("GPT-2 runs on the HIP kernels only (GPU tensor required)")
        r, l = tokens.shape
        km = None if attention_mask is None else attention_mask.to(torch.uint8).contiguous()
        h = ops.gpt2_embed(tokens, self.P("transformer.wte.weight"), self.P("transformer.wpe.weight"))
        for i in range(self.n_layer):
            h = self._block(i, h, lambda _i, qkv: ops.attn_causal(qkv, km, r, l, self.n_head))
        return self._head(h).view(r, l, -1)

    @torch.no_grad()
    def forward_step(self, last_tokens, state: KVCacheState, key_mask=None, max_len=None):
        """One cached step: last_tokens i64 [rows] at position state.len -> logits [rows, V]."""
        rows = last_tokens.shape[0]
        dh = self.d_model // self.n_head
        if state.k is None:
            lmax = max_len or self.n_positions
            shape = (self.n_layer, rows, self.n_head, lmax, dh)
            dev = last_tokens.device
            state.k = torch.zeros(shape, dtype=torch.float32, device=dev)
            state.v = torch.zeros(shape, dtype=torch.float32, device=dev)
        t = state.len
        h = ops.gpt2_embed(last_tokens.view(rows, 1), self.P("transformer.wte.weight"),
                           self.P("transformer.wpe.weight"), pos0=t)
        if 16 < rows <= 64 and self.d_model % 128 == 0 and (self.d_model // self.n_head) in (16, 32, 64):
            return self._forward_step_packed(h, rows, state, key_mask, t)
        for i in range(self.n_layer):
            h = self._block(i, h, lambda li, qkv: ops.attn_decode(qkv, state.k[li][:rows], state.v[li][:rows],
                                                                  key_mask, t, ancestry=state.anc))
        state.len = t + 1
        return self._head(h)

    def _forward_step_packed(self, h, rows, state, key_mask, t):
        """The cached step for 17..64 rows (sentences x beams): every GEMM operand fragment-major --
        weights packed once, activations written in that layout by their producers (layernorm,
        attention, the gelu epilogue) -- so each MFMA fragment load is one contiguous KB."""
        d, v = self.d_model, self.P("transformer.wte.weight").shape[0]
        for i in range(self.n_layer):
            q = f"transformer.h.{i}."
            a = ops.layernorm_fwd_packed(h, self.P(q + "ln_1.weight"), self.P(q + "ln_1.bias"), 1e-5)
            qkv = ops.gemm_nt_packed(a, self._wp(q + "attn.c_attn.weight"), rows, 3 * d, d,
                                     b=self.P(q + "attn.c_attn.bias"))
            o = ops.attn_decode(qkv, state.k[i][:rows], state.v[i][:rows], key_mask, t, ancestry=state.anc,
                                out_packed=True)
            h = ops.gemm_nt_packed(o, self._wp(q + "attn.c_proj.weight"), rows, d, d,
                                   b=self.P(q + "attn.c_proj.bias"), res=h)
            m = ops.layernorm_fwd_packed(h, self.P(q + "ln_2.weight"), self.P(q + "ln_2.bias"), 1e-5)
            f = ops.gemm_nt_packed(m, self._wp(q + "mlp.c_fc.weight"), rows, 4 * d, d,
                                   b=self.P(q + "mlp.c_fc.bias"), act=ops.ACT_GELU_NEW, y_packed=True)
            h = ops.gemm_nt_packed(f, self._wp(q + "mlp.c_proj.weight"), rows, d, 4 * d,
                                   b=self.P(q + "mlp.c_proj.bias"), res=h)
        state.len = t + 1
        hf = ops.layernorm_fwd_packed(h, self.P("transformer.ln_f.weight"), self.P("transformer.ln_f.bias"), 1e-5)
        return ops.gemm_nt_packed(hf, self._wp("transformer.wte.weight", transposed=False), rows, v, d)

    @torch.no_grad()
    def generate_greedy(self, input_ids, max_length, pad_token_id, eos_token_id, sync_every=8):
        """huggingface `generate(num_beams=1, do_sample=False, use_cache=True)` as `Simple_GPT2.forward_gen`
        calls it (mdl_sf_base.py:494-503, 577-585): argmax of the newest position on the cached decode
        step, rows that have emitted eos receive pad from then on, and the result ends at the step where
        the last row finished (or at max_length).  The all-finished test reads the device only every
        `sync_every` steps; columns generated past the stopping step are cut off afterwards, so the result
        is the one of a loop testing after every token.  -> i64 [rows, <= max_length]"""
        rows, plen = input_ids.shape
        if not input_ids.is_cuda:
            raise ops._lib.VsError("GPT-2 runs on the HIP kernels only (GPU tensor required)")
        if not 1 <= plen <= max_length <= self.n_positions:
            raise ops._lib.VsError(f"prompt {plen} / max_length {max_length} / n_positions {self.n_positions}")
        state = KVCacheState()
        out = torch.full((rows, max_length), pad_token_id, dtype=torch.int64, device=input_ids.device)
        out[:, :plen] = input_ids
        unfinished = torch.ones(rows, dtype=torch.bool, device=input_ids.device)
        alive = torch.ones(max_length, dtype=torch.bool, device=input_ids.device)  # any row unfinished after col t
        logits = None
        for t in range(plen):  # the prompt, one cached position at a time (prompts here are one token)
            logits = self.forward_step(input_ids[:, t].contiguous(), state, max_len=max_length)
        end = max_length
        for t in range(plen, max_length):
            _, nxt = ops.softmax_topk(logits, 1)
            add = torch.where(unfinished, nxt[:, 0], torch.full_like(nxt[:, 0], pad_token_id))
            out[:, t] = add
            unfinished = unfinished & add.ne(eos_token_id)
            alive[t] = unfinished.any()
            if (t - plen) % sync_every == sync_every - 1 or t == max_length - 1:
                dead = (~alive[plen:t + 1]).nonzero()
                if dead.numel():
                    end = plen + int(dead[0]) + 1
                    break
            if t + 1 < max_length:
                logits = self.forward_step(add, state, max_len=max_length)
        return out[:, :end].contiguous()

    # ---- training (teacher-forced pass with saved activations + manual backward) ------------
    def forward_train(self, tokens, attention_mask=None):
        """Like forward_logits, keeping what the backward needs (fp32, no recompute except the
        attention probabilities)."""
        r, l = tokens.shape
        km = None if attention_mask is None else attention_mask.to(torch.uint8).contiguous()
        h = ops.gpt2_embed(tokens, self.P("transformer.wte.weight"), self.P("transformer.wpe.weight"))
        layers = []
        for i in range(self.n_layer):
            q = f"transformer.h.{i}."
            a, mean1, rstd1 = ops.add_layernorm_fwd(h, None, self.P(q + "ln_1.weight"),
                                                    self.P(q + "ln_1.bias"), 1e-5)
            qkv = ops.gemm_nt(a, self._w(q + "attn.c_attn.weight"), self.P(q + "attn.c_attn.bias"))
            o = ops.attn_causal(qkv, km, r, l, self.n_head)
            h_mid = ops.gemm_nt(o, self._w(q + "attn.c_proj.weight"), self.P(q + "attn.c_proj.bias"), res=h)
            m, mean2, rstd2 = ops.add_layernorm_fwd(h_mid, None, self.P(q + "ln_2.weight"),
                                                    self.P(q + "ln_2.bias"), 1e-5)
            f_pre = ops.gemm_nt(m, self._w(q + "mlp.c_fc.weight"), self.P(q + "mlp.c_fc.bias"))
            f = ops.gelu_new_fwd(f_pre)
            h_out = ops.gemm_nt(f, self._w(q + "mlp.c_proj.weight"), self.P(q + "mlp.c_proj.bias"), res=h_mid)
            layers.append((h, a, mean1, rstd1, qkv, o, h_mid, m, mean2, rstd2, f_pre, f))
            h = h_out
        hf, meanf, rstdf = ops.add_layernorm_fwd(h, None, self.P("transformer.ln_f.weight"),
                                                 self.P("transformer.ln_f.bias"), 1e-5)
        logits = ops.gemm_nt(hf, self.P("transformer.wte.weight"))
        self._saved = (tokens, km, r, l, layers, h, hf, meanf, rstdf)
        return logits.view(r, l, -1)

    def _grad(self, name):
        p = self.P(name)
        if p.grad is None:
            p.grad = torch.empty_like(p)
        return p.grad

    def _dense_bwd(self, name, x, dy, need_dx=True):
        """y = x @ W + b with the Conv1D parameter W [in, out]: dW = x^T dy, db = colsum(dy),
        dx = dy @ W^T (the parameter itself is the K-contiguous operand of that product)."""
        ops.gemm_nt(ops.transpose_f32(x), ops.transpose_f32(dy), out=self._grad(name + ".weight"))
        ops.colsum_f32(dy, out=self._grad(name + ".bias"))
        return ops.gemm_nt(dy, self.P(name + ".weight")) if need_dx else None

    @torch.no_grad()
    def backward(self, dlogits):
        """dlogits [R*L, V] -> .grad of every parameter (overwrite semantics, like the trunk)."""
        tokens, km, r, l, layers, h_last, hf, meanf, rstdf = self._saved
        self._saved = None
        wte = self.P("transformer.wte.weight")
        dlogits = dlogits.reshape(r * l, -1)
        # tied lm_head: logits = hf @ wte^T
        dwte = self._grad("transformer.wte.weight")
        ops.gemm_nt(ops.transpose_f32(dlogits), ops.transpose_f32(hf), out=dwte)
        dhf = ops.gemm_nt(dlogits, ops.transpose_f32(wte))
        dh, _, _, _ = ops.add_layernorm_bwd(dhf, h_last, None, self.P("transformer.ln_f.weight"), meanf,
                                            rstdf, dg_out=self._grad("transformer.ln_f.weight"),
                                            db_out=self._grad("transformer.ln_f.bias"))
        for i in reversed(range(self.n_layer)):
            q = f"transformer.h.{i}."
            h_in, a, mean1, rstd1, qkv, o, h_mid, m, mean2, rstd2, f_pre, f = layers[i]
            df = self._dense_bwd(q + "mlp.c_proj", f, dh)
            df_pre = ops.gelu_new_bwd(df, f_pre)
            dm = self._dense_bwd(q + "mlp.c_fc", m, df_pre)
            dmid_ln, _, _, _ = ops.add_layernorm_bwd(dm, h_mid, None, self.P(q + "ln_2.weight"), mean2, rstd2,
                                                     dg_out=self._grad(q + "ln_2.weight"),
                                                     db_out=self._grad(q + "ln_2.bias"))
            d_mid = ops.add_f32(dh, dmid_ln)
            do = self._dense_bwd(q + "attn.c_proj", o, d_mid)
            dqkv = ops.attn_causal_bwd(qkv, km, do, r, l, self.n_head)
            da = self._dense_bwd(q + "attn.c_attn", a, dqkv)
            din_ln, _, _, _ = ops.add_layernorm_bwd(da, h_in, None, self.P(q + "ln_1.weight"), mean1, rstd1,
                                                    dg_out=self._grad(q + "ln_1.weight"),
                                                    db_out=self._grad(q + "ln_1.bias"))
            dh = ops.add_f32(d_mid, din_ln)
            layers[i] = None
        dwpe = self._grad("transformer.wpe.weight")
        dwpe.zero_()
        ops.gpt2_embed_bwd(tokens, dh, dwte, dwpe)  # adds the embedding rows onto the lm_head gradient
        self._drop_wt()  # the optimizer is about to change the parameters

    def take_train_state(self):
        """The activations `forward_train` saved for `backward`, detached from the module (the autograd node keeps
        them: several forwards may precede a backward)."""
        st, self._saved = self._saved, None
        return st

    def put_train_state(self, st):
        self._saved = st

    def reorder_state(self, state: KVCacheState, new_order):
        """fairseq reorder_incremental_state: row r of the cache becomes old row new_order[r]."""
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
        state.rows = rows


class _GPT2TrainFn(torch.autograd.Function):
    """One autograd node for the whole language model (manual HIP backward)."""

    @staticmethod
    def forward(ctx, model, tokens, mask, _tick):
        ctx.model = model
        out = model.forward_train(tokens, mask)
        # the saved activations of THIS forward travel with the node: a second forward before the backward
        # (two losses, gradient accumulation) must not replace them
        ctx.saved_state = model.take_train_state() if hasattr(model, "take_train_state") else None
        return out

    @staticmethod
    def backward(ctx, dlogits):
        if ctx.saved_state is not None:
            ctx.model.put_train_state(ctx.saved_state)
        ctx.model.backward(dlogits.contiguous())
        return None, None, None, None


class _XentIgnoreFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits2d, labels, ignore_index):
        loss, pair = ops.xent_ignore(logits2d, labels, ignore_index)
        ctx.save_for_backward(logits2d, labels, pair)
        ctx.ignore_index = ignore_index
        return loss.clone()

    @staticmethod
    def backward(ctx, go):
        logits2d, labels, pair = ctx.saved_tensors
        return ops.xent_ignore_grad(logits2d, labels, pair, ctx.ignore_index, go), None, None  # go stays on the device


class HuggingFaceGPT2Decoder(nn.Module):
    """hf_gpt2_fseq.py:124-215.  `args` is the full config (`args.mdl.gpt2_mdl_name`), `dictionary`
    the tokenizer object (`pad()`, `eos()`, `__len__`)."""

    # `forward` never reads encoder_out (hf_gpt2_fseq.py:165-203 does not either): decode steps may be
    # captured in hipGraphs without pinning the encoder output's address
    uses_encoder_out = False

    def __init__(self, args, dictionary, state_dict=None):
        super().__init__()
        self.dictionary = dictionary
        dims = GPT2_DIMS[args.mdl.gpt2_mdl_name]
        self.model = GPT2LMHeadModelHip(*dims)
        if state_dict is not None:
            self.model.load_state_dict(state_dict)
        self.voc_size = len(dictionary)
        self.model.resize_token_embeddings(self.voc_size)
        self.pad_idx = dictionary.pad()

    def forward(self, prev_output_tokens, src_lengths=None, incremental_state=None, encoder_out=None):
        if incremental_state:  # cached single-position step (see module docstring)
            if incremental_state.len != prev_output_tokens.size(1) - 1:
                raise ops._lib.VsError("incremental state out of step with prev_output_tokens")
            logits = self.model.forward_step(prev_output_tokens[:, -1].contiguous(), incremental_state,
                                             max_len=getattr(incremental_state, "max_len", None))
            return (logits.unsqueeze(1),)
        features_mask = prev_output_tokens.ne(self.pad_idx)  # don't attend to padding symbols (:185)
        if self.training and torch.is_grad_enabled():
            tick = torch.zeros(1, device=prev_output_tokens.device, requires_grad=True)
            return (_GPT2TrainFn.apply(self.model, prev_output_tokens, features_mask, tick),)
        return (self.model.forward_logits(prev_output_tokens, features_mask),)

    def reorder_incremental_state(self, incremental_state, new_order):
        if incremental_state:
            self.model.reorder_state(incremental_state, new_order)

    def max_positions(self):
        return self.model.config.n_positions - 1

    def max_decoder_positions(self):
        return self.model.config.n_positions - 1


def lm_loss(logits, tokens, pad_index):
    """`Simple_TxDec.forward` (`mdl_sf_base.py:653-667`): CE(logits[:, :-1], tokens[:, 1:]),
    ignore_index = pad, mean over the counted tokens."""
    r, l, v = logits.shape
    labels = torch.full((r, l), pad_index, dtype=torch.int64, device=tokens.device)
    labels[:, :-1] = tokens[:, 1:]
    if logits.requires_grad:
        return _XentIgnoreFn.apply(logits.reshape(r * l, v), labels.reshape(-1), pad_index)
    loss, _ = ops.xent_ignore(logits.reshape(r * l, v), labels.reshape(-1), pad_index)
    return loss
