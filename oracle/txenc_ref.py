"""
ORACLE (test infrastructure, not product code) -- CPU restatement of the
reference's self-contained post-LN transformer encoder
(`/root/reference/utils/transformer_code.py:21-124`), the arithmetic behind
`TxEncoderNew` (`vidsitu_code/mdl_sf_base.py:341-381`).

PARITY PINNED: tests/golden/txenc_*.npz were produced by importing the
reference module itself in the build container
(tests/golden/gen_txenc_golden.py) and this restatement is checked against them
in tests/test_oracle_txenc.py.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may
import this package.

Weights are passed as a flat dict using the reference's state_dict names:
  layers.{i}.selfattn.layer.{wq,wk,wv,wo}.weight              [D, D] (no bias)
  layers.{i}.selfattn.layernorm.{weight,bias}                 [D]
  layers.{i}.feedforward.layer.linear{1,2}.{weight,bias}
  layers.{i}.feedforward.layernorm.{weight,bias}
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def make_weights(d_model, d_hidden, n_layers, seed, dtype=np.float32):
    """Deterministic weights from numpy's legacy RandomState (stable across
    numpy versions) so fixtures need not store the 6.3 M params per layer."""
    rs = np.random.RandomState(seed)
    w = {}

    def lin(name, dout, din, bias):
        w[name + ".weight"] = (rs.standard_normal((dout, din)) / math.sqrt(din)).astype(dtype)
        if bias:
            w[name + ".bias"] = (0.1 * rs.standard_normal(dout)).astype(dtype)

    def ln(name, d):
        w[name + ".weight"] = (1.0 + 0.1 * rs.standard_normal(d)).astype(dtype)
        w[name + ".bias"] = (0.1 * rs.standard_normal(d)).astype(dtype)

    for i in range(n_layers):
        p = f"layers.{i}."
        for nm in ("wq", "wk", "wv", "wo"):
            lin(p + "selfattn.layer." + nm, d_model, d_model, False)
        ln(p + "selfattn.layernorm", d_model)
        lin(p + "feedforward.layer.linear1", d_hidden, d_model, True)
        lin(p + "feedforward.layer.linear2", d_model, d_hidden, True)
        ln(p + "feedforward.layernorm", d_model)
    return w


def encoder_layer(x, w, p, n_heads, eps=1e-5):
    """transformer_code.py:82-93 (EncoderLayer) in eval mode (dropout = id)."""
    d = x.shape[-1]
    q = x @ w[p + "selfattn.layer.wq.weight"].T  # :61
    k = x @ w[p + "selfattn.layer.wk.weight"].T
    v = x @ w[p + "selfattn.layer.wv.weight"].T
    dh = d // n_heads
    heads = []
    for h in range(n_heads):  # :63-68 chunk(n_heads, -1), per-head attention
        sl = slice(h * dh, (h + 1) * dh)
        # :36,48 -- scale is sqrt(d_key) with d_key = d_model, NOT head_dim
        s = (q[..., sl] @ k[..., sl].transpose(-1, -2)) / math.sqrt(d)
        heads.append(F.softmax(s, dim=-1) @ v[..., sl])
    a = torch.cat(heads, -1) @ w[p + "selfattn.layer.wo.weight"].T
    x1 = F.layer_norm(  # :30 LN(x + layer(x))
        x + a, (d,), w[p + "selfattn.layernorm.weight"], w[p + "selfattn.layernorm.bias"], eps
    )
    h1 = F.relu(
        x1 @ w[p + "feedforward.layer.linear1.weight"].T + w[p + "feedforward.layer.linear1.bias"]
    )
    f = h1 @ w[p + "feedforward.layer.linear2.weight"].T + w[p + "feedforward.layer.linear2.bias"]
    return F.layer_norm(
        x1 + f, (d,), w[p + "feedforward.layernorm.weight"], w[p + "feedforward.layernorm.bias"], eps
    )


def encoder_forward(x, weights, n_layers, n_heads, dtype=torch.float32):
    """Encoder.forward(x)[-1] (transformer_code.py:109-124), eval mode.
    x: [B, L, D] array/tensor -> tensor [B, L, D]."""
    w = {k: torch.as_tensor(v).to(dtype) for k, v in weights.items()}
    x = torch.as_tensor(x).to(dtype)
    for i in range(n_layers):
        x = encoder_layer(x, w, f"layers.{i}.", n_heads)
    return x
