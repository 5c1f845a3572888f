"""ORACLE (test infrastructure only -- never imported by the product path).

CPU fp32 restatement (plain torch ops, autograd gives the gradients) of fairseq's `TransformerDecoder`
as `TxDecoderReal` configures it (`vidsitu_code/mdl_sf_base.py:435-446`, `configs/vsitu_tx_cfgs/
transformer.yaml`: 3 post-norm layers, d 1024, 8 heads, ffn 2048 relu, sinusoidal positions,
embed scale sqrt(d), cross attention over the encoder output, `project_out_dim` 1024 -> 512 and an
untied `output_projection` 512 -> V, both without bias).  The arithmetic lives in the un-vendored
fairseq fork `TheShadow29/fairseq@fseq_master_13Oct20` (`.gitmodules`), modules
`fairseq/models/transformer.py` (TransformerDecoder.extract_features), `modules/transformer_layer.py`
(TransformerDecoderLayer, post-norm branch), `modules/multihead_attention.py` (q scaled by
head_dim**-0.5 after the projection, -inf masks, fp32 softmax), `modules/sinusoidal_positional_
embedding.py` (sin | cos halves, positions padding_idx + 1 + t, zero row at padding_idx),
`utils.make_positions`.

PARITY UNPINNED: fairseq is not installed and the reference holds no test or golden for this module.
Cross-checked in `tests/test_oracle_txdec.py` against `torch.nn.TransformerDecoderLayer` (an
independent implementation of the same post-norm layer) with the weights mapped.
"""
import math

import torch
import torch.nn.functional as F


def param_names(n_layer):
    names = ["embed_tokens.weight"]
    for i in range(n_layer):
        q = f"layers.{i}."
        for att in ("self_attn", "encoder_attn"):
            for pr in ("q_proj", "k_proj", "v_proj", "out_proj"):
                names += [q + f"{att}.{pr}.weight", q + f"{att}.{pr}.bias"]
            names += [q + f"{att}_layer_norm.weight", q + f"{att}_layer_norm.bias"]
        names += [q + "fc1.weight", q + "fc1.bias", q + "fc2.weight", q + "fc2.bias",
                  q + "final_layer_norm.weight", q + "final_layer_norm.bias"]
    return names + ["project_out_dim.weight", "output_projection.weight"]


def make_weights(vocab, d, ffn, n_layer, out_dim, pad, seed=0):
    g = torch.Generator().manual_seed(seed)
    w = {}
    for n in param_names(n_layer):
        if n == "embed_tokens.weight":
            t = torch.randn(vocab, d, generator=g) * d ** -0.5
            t[pad] = 0
        elif n.endswith("layer_norm.weight"):
            t = 1.0 + 0.1 * torch.randn(d, generator=g)
        elif n.endswith("layer_norm.bias") or n.endswith(".bias"):
            dim = ffn if n.endswith("fc1.bias") else d
            t = 0.05 * torch.randn(dim, generator=g)
        elif n.endswith("fc1.weight"):
            t = torch.randn(ffn, d, generator=g) * d ** -0.5
        elif n.endswith("fc2.weight"):
            t = torch.randn(d, ffn, generator=g) * ffn ** -0.5
        elif n == "project_out_dim.weight":
            t = torch.randn(out_dim, d, generator=g) * d ** -0.5
        elif n == "output_projection.weight":
            t = torch.randn(vocab, out_dim, generator=g) * out_dim ** -0.5
        else:
            t = torch.randn(d, d, generator=g) * d ** -0.5
        w[n] = t
    return w


def sinusoidal_rows(positions, d):
    """Rows `positions` (int64 tensor) of fairseq's SinusoidalPositionalEmbedding.get_embedding table,
    computed in fp32 in the table's operation order; the padding row is zeroed by the caller."""
    half = d // 2
    freq = torch.exp(torch.arange(half, dtype=torch.float) * -(math.log(10000) / (half - 1)))
    ang = positions.to(torch.float).unsqueeze(-1) * freq
    emb = torch.cat([torch.sin(ang), torch.cos(ang)], dim=-1)
    if d % 2 == 1:
        emb = torch.cat([emb, torch.zeros_like(emb[..., :1])], dim=-1)
    return emb


def make_positions(tokens, pad):
    mask = tokens.ne(pad).int()
    return (torch.cumsum(mask, dim=1).type_as(mask) * mask).long() + pad


def _mha(x_q, x_kv, w, pre, n_head, attn_mask=None, key_padding_mask=None):
    """fairseq MultiheadAttention.forward on batch-first tensors [B, L, D] / [B, S, D]."""
    b, l, d = x_q.shape
    s = x_kv.shape[1]
    dh = d // n_head
    q = F.linear(x_q, w[pre + "q_proj.weight"], w[pre + "q_proj.bias"]) * dh ** -0.5
    k = F.linear(x_kv, w[pre + "k_proj.weight"], w[pre + "k_proj.bias"])
    v = F.linear(x_kv, w[pre + "v_proj.weight"], w[pre + "v_proj.bias"])
    q = q.view(b, l, n_head, dh).transpose(1, 2)
    k = k.view(b, s, n_head, dh).transpose(1, 2)
    v = v.view(b, s, n_head, dh).transpose(1, 2)
    a = q @ k.transpose(-1, -2)
    if attn_mask is not None:
        a = a + attn_mask
    if key_padding_mask is not None:
        a = a.masked_fill(key_padding_mask[:, None, None, :], float("-inf"))
    p = torch.softmax(a.float(), dim=-1)
    o = (p @ v).transpose(1, 2).reshape(b, l, d)
    return F.linear(o, w[pre + "out_proj.weight"], w[pre + "out_proj.bias"])


def forward(w, tokens, enc, pad, n_head, n_layer):
    """tokens i64 [B, L]; enc f32 [S, B, D] (fairseq's T x B x C encoder output) or None
    -> logits f32 [B, L, V] (eval mode: no dropout)."""
    d = w["embed_tokens.weight"].shape[1]
    b, l = tokens.shape
    pos = make_positions(tokens, pad)
    pe = sinusoidal_rows(pos, d)
    pe = pe * pos.ne(pad).unsqueeze(-1)  # the table's padding row is zero
    x = math.sqrt(d) * F.embedding(tokens, w["embed_tokens.weight"]) + pe
    pad_mask = tokens.eq(pad)
    kpm = pad_mask if bool(pad_mask.any()) else None
    causal = torch.triu(torch.full((l, l), float("-inf")), 1)
    enc_b = None if enc is None else enc.transpose(0, 1)
    for i in range(n_layer):
        q = f"layers.{i}."
        x = F.layer_norm(x + _mha(x, x, w, q + "self_attn.", n_head, causal, kpm), (d,),
                         w[q + "self_attn_layer_norm.weight"], w[q + "self_attn_layer_norm.bias"])
        if enc_b is not None:
            x = F.layer_norm(x + _mha(x, enc_b, w, q + "encoder_attn.", n_head), (d,),
                             w[q + "encoder_attn_layer_norm.weight"], w[q + "encoder_attn_layer_norm.bias"])
        f = F.linear(F.relu(F.linear(x, w[q + "fc1.weight"], w[q + "fc1.bias"])), w[q + "fc2.weight"],
                     w[q + "fc2.bias"])
        x = F.layer_norm(x + f, (d,), w[q + "final_layer_norm.weight"], w[q + "final_layer_norm.bias"])
    x = F.linear(x, w["project_out_dim.weight"])
    return F.linear(x, w["output_projection.weight"])


def lm_loss(logits, tokens, pad):
    """Simple_TxDec.forward (mdl_sf_base.py:653-667): CE(logits[:, :-1], tokens[:, 1:]), ignore pad."""
    v = logits.shape[-1]
    return F.cross_entropy(logits[:, :-1].reshape(-1, v), tokens[:, 1:].reshape(-1), ignore_index=pad)


# ---- fairseq TransformerEncoder as TxEncoderOld calls it (mdl_sf_base.py:246-338): the video features are
#      the token embeddings, `src_tokens` (their first feature channel, a float tensor) only feeds the
#      position / padding rules; post-norm layers without a final norm ----------------------------------
def encoder_param_names(n_layer):
    names = []
    for i in range(n_layer):
        q = f"layers.{i}."
        for pr in ("q_proj", "k_proj", "v_proj", "out_proj"):
            names += [q + f"self_attn.{pr}.weight", q + f"self_attn.{pr}.bias"]
        names += [q + "self_attn_layer_norm.weight", q + "self_attn_layer_norm.bias", q + "fc1.weight",
                  q + "fc1.bias", q + "fc2.weight", q + "fc2.bias", q + "final_layer_norm.weight",
                  q + "final_layer_norm.bias"]
    return names


def make_encoder_weights(d, ffn, n_layer, seed=0):
    g = torch.Generator().manual_seed(seed)
    w = {}
    for n in encoder_param_names(n_layer):
        if n.endswith("layer_norm.weight"):
            t = 1.0 + 0.1 * torch.randn(d, generator=g)
        elif n.endswith(".bias"):
            t = 0.05 * torch.randn(ffn if n.endswith("fc1.bias") else d, generator=g)
        elif n.endswith("fc1.weight"):
            t = torch.randn(ffn, d, generator=g) * d ** -0.5
        elif n.endswith("fc2.weight"):
            t = torch.randn(d, ffn, generator=g) * ffn ** -0.5
        else:
            t = torch.randn(d, d, generator=g) * d ** -0.5
        w[n] = t
    return w


def encoder_forward(w, token_embeddings, src_tokens, pad, n_head, n_layer):
    """token_embeddings f32 [B, L, D], src_tokens [B, L] -> encoder_out [L, B, D] (eval mode)."""
    d = token_embeddings.shape[-1]
    pos = make_positions(src_tokens, pad)
    pe = sinusoidal_rows(pos, d) * pos.ne(pad).unsqueeze(-1)
    x = math.sqrt(d) * token_embeddings + pe
    pad_mask = src_tokens.eq(pad)
    kpm = pad_mask if bool(pad_mask.any()) else None
    for i in range(n_layer):
        q = f"layers.{i}."
        x = F.layer_norm(x + _mha(x, x, w, q + "self_attn.", n_head, None, kpm), (d,),
                         w[q + "self_attn_layer_norm.weight"], w[q + "self_attn_layer_norm.bias"])
        f = F.linear(F.relu(F.linear(x, w[q + "fc1.weight"], w[q + "fc1.bias"])), w[q + "fc2.weight"],
                     w[q + "fc2.bias"])
        x = F.layer_norm(x + f, (d,), w[q + "final_layer_norm.weight"], w[q + "final_layer_norm.bias"])
    return x.transpose(0, 1)


def encoder_conc_forward(w, token_embeddings, src_tokens, pad, n_head, n_layer):
    """`TxEncoderNew_Conc.forward` (`mdl_sf_base.py:402-420`): cat(features, encoder output) ->
    Linear-ReLU-Linear (`orig_tx_out_comb.0/.2`) -> [L, B, D]."""
    enc = encoder_forward(w, token_embeddings, src_tokens, pad, n_head, n_layer).transpose(0, 1)
    h = F.relu(F.linear(torch.cat([token_embeddings, enc], dim=-1), w["orig_tx_out_comb.0.weight"],
                        w["orig_tx_out_comb.0.bias"]))
    return F.linear(h, w["orig_tx_out_comb.2.weight"], w["orig_tx_out_comb.2.bias"]).transpose(0, 1)
