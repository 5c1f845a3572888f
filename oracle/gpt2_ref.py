"""ORACLE (test infrastructure, never the product path): numpy fp32 restatement of the GPT-2
language model the reference decodes with (`vidsitu_code/hf_gpt2_fseq.py:124-215`:
`HuggingFaceGPT2Decoder.extract_features` -> `self.model.transformer(input_ids, attention_mask=
tokens != pad)` -> `self.model.lm_head`).  The arithmetic lives in the third-party dependency
**huggingface/transformers, pinned `transformers==3.3.1`** (`vsitu_pyt_env.yml:320`),
`modeling_gpt2.py` (`GPT2Model`, `Block`, `Attention`, `MLP`, `Conv1D`), absent from /root/reference.
Published algorithm restated here:
  h0 = wte[tok] + wpe[arange(L)]                     (default position_ids: not mask-aware)
  per block:  a = LN1(h);  q,k,v = split(a @ Wqkv + b);  heads of d_head = D / n_head;
              w = q k^T / sqrt(d_head);  w = where(causal, w, -1e4);  w += (1 - mask_k) * -1e4;
              p = softmax(w);  h += merge(p v) @ Wproj + b;
              m = LN2(h);  h += gelu_new(m @ Wfc + b) @ Wproj2 + b
  logits = LNf(h) @ wte^T                            (tied, bias-free lm_head)
  gelu_new(x) = 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3))),  LN eps 1e-5.
PINNED: `tests/golden/gpt2_*.npz` hold logits produced HERE by the installed transformers
`GPT2LMHeadModel` (same published algorithm; masks use finfo.min instead of -1e4, identical after
softmax in fp32) on seeded weights (`tests/golden/gen_gpt2_golden.py`); `tests/test_oracle_gpt2.py`
checks this restatement against them.
"""
import math

import numpy as np


def gpt2_dims(name):
    return {"gpt2-medium": dict(n_layer=24, d=1024, n_head=16, n_pos=1024),
            "gpt2": dict(n_layer=12, d=768, n_head=12, n_pos=1024)}[name]


def make_weights(vocab, n_pos, d, n_layer, seed, std=0.02, ln_jitter=0.1):
    """Seeded weights with the HF state-dict key names (`transformer.*`, Conv1D weights [in, out]).
    LayerNorm gains / all biases are jittered so that a mistake in any of them shows."""
    rs = np.random.RandomState(seed)
    f32 = np.float32
    w = {"transformer.wte.weight": (rs.randn(vocab, d) * std * 5).astype(f32),
         "transformer.wpe.weight": (rs.randn(n_pos, d) * std * 5).astype(f32)}
    for i in range(n_layer):
        p = f"transformer.h.{i}."
        for ln in ("ln_1", "ln_2"):
            w[p + ln + ".weight"] = (1.0 + ln_jitter * rs.randn(d)).astype(f32)
            w[p + ln + ".bias"] = (ln_jitter * rs.randn(d)).astype(f32)
        w[p + "attn.c_attn.weight"] = (rs.randn(d, 3 * d) * std * 3).astype(f32)
        w[p + "attn.c_attn.bias"] = (rs.randn(3 * d) * std).astype(f32)
        w[p + "attn.c_proj.weight"] = (rs.randn(d, d) * std * 3).astype(f32)
        w[p + "attn.c_proj.bias"] = (rs.randn(d) * std).astype(f32)
        w[p + "mlp.c_fc.weight"] = (rs.randn(d, 4 * d) * std * 3).astype(f32)
        w[p + "mlp.c_fc.bias"] = (rs.randn(4 * d) * std).astype(f32)
        w[p + "mlp.c_proj.weight"] = (rs.randn(4 * d, d) * std * 3).astype(f32)
        w[p + "mlp.c_proj.bias"] = (rs.randn(d) * std).astype(f32)
    w["transformer.ln_f.weight"] = (1.0 + ln_jitter * rs.randn(d)).astype(f32)
    w["transformer.ln_f.bias"] = (ln_jitter * rs.randn(d)).astype(f32)
    return w


def layer_norm(x, g, b, eps=1e-5):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * g + b


def gelu_new(x):
    return 0.5 * x * (1.0 + np.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3)))


def forward(w, tokens, attention_mask, n_head, n_layer=None, return_hidden=False):
    """tokens i64 [R, L], attention_mask {0,1} [R, L] -> logits f32 [R, L, V]."""
    tokens = np.asarray(tokens)
    R, L = tokens.shape
    d = w["transformer.wte.weight"].shape[1]
    dh = d // n_head
    if n_layer is None:
        n_layer = 1 + max(int(k.split(".")[2]) for k in w if k.startswith("transformer.h."))
    h = w["transformer.wte.weight"][tokens] + w["transformer.wpe.weight"][np.arange(L)][None]
    causal = np.tril(np.ones((L, L), dtype=bool))
    kmask = (1.0 - np.asarray(attention_mask, dtype=np.float32))[:, None, None, :] * -1e4
    for i in range(n_layer):
        p = f"transformer.h.{i}."
        a = layer_norm(h, w[p + "ln_1.weight"], w[p + "ln_1.bias"])
        qkv = a @ w[p + "attn.c_attn.weight"] + w[p + "attn.c_attn.bias"]
        q, k, v = np.split(qkv, 3, axis=-1)
        sh = lambda t: t.reshape(R, L, n_head, dh).transpose(0, 2, 1, 3)
        q, k, v = sh(q), sh(k), sh(v)
        s = (q @ k.transpose(0, 1, 3, 2)) / np.float32(math.sqrt(dh))
        s = np.where(causal[None, None], s, np.float32(-1e4)) + kmask
        s = s - s.max(-1, keepdims=True)
        pr = np.exp(s)
        pr = pr / pr.sum(-1, keepdims=True)
        o = (pr @ v).transpose(0, 2, 1, 3).reshape(R, L, d)
        h = h + o @ w[p + "attn.c_proj.weight"] + w[p + "attn.c_proj.bias"]
        m = layer_norm(h, w[p + "ln_2.weight"], w[p + "ln_2.bias"])
        f = gelu_new(m @ w[p + "mlp.c_fc.weight"] + w[p + "mlp.c_fc.bias"])
        h = h + f @ w[p + "mlp.c_proj.weight"] + w[p + "mlp.c_proj.bias"]
    hf = layer_norm(h, w["transformer.ln_f.weight"], w["transformer.ln_f.bias"])
    logits = (hf @ w["transformer.wte.weight"].T).astype(np.float32)
    return (logits, hf.astype(np.float32)) if return_hidden else logits


def lm_loss(logits, tokens, pad):
    """`Simple_TxDec.forward` (`mdl_sf_base.py:653-667`): shift by one, mean CE, ignore_index=pad."""
    lg = logits[:, :-1].reshape(-1, logits.shape[-1]).astype(np.float64)
    lb = np.asarray(tokens)[:, 1:].reshape(-1)
    lse = np.log(np.exp(lg - lg.max(-1, keepdims=True)).sum(-1)) + lg.max(-1)
    nll = lse - lg[np.arange(len(lb)), lb]
    keep = lb != pad
    return float(nll[keep].mean())


def greedy_generate(w, input_ids, max_length, pad, eos, n_head):
    """huggingface `generate(num_beams=1, do_sample=False)` as `Simple_GPT2.forward_gen` /
    `Simple_GPT2_New.forward_gen` call it (`mdl_sf_base.py:494-503, 577-585`): argmax of the last position,
    rows that have emitted `eos` receive `pad` from then on, the loop ends when every row has or at
    `max_length` tokens.  Restates transformers' published greedy loop (third-party, the reference pins
    3.3.1: generation_utils.py `_generate_no_beam_search`); pinned by tests/golden/greedy_gpt2_tiny.npz.
    -> i64 [R, <= max_length]."""
    ids = np.asarray(input_ids, dtype=np.int64)
    unfinished = np.ones(ids.shape[0], dtype=bool)
    while ids.shape[1] < max_length:
        logits = forward(w, ids, np.ones_like(ids), n_head)[:, -1]
        nxt = logits.argmax(-1)
        add = np.where(unfinished, nxt, pad)
        ids = np.concatenate([ids, add[:, None]], axis=1)
        unfinished &= add != eos
        if not unfinished.any():
            break
    return ids
