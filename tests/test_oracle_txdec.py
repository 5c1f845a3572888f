"""The fairseq-TransformerDecoder oracle (`oracle/txdec_ref.py`, parity unpinned: fairseq is absent)
cross-checked against an independent implementation of the same post-norm decoder layer,
`torch.nn.TransformerDecoderLayer`, with the weights mapped; plus the position rules."""
import math

import torch

from oracle import txdec_ref


def test_positions_and_sinusoid_table():
    pad = 7
    toks = torch.tensor([[3, 4, 5, pad, pad], [1, 2, 3, 4, 5]])
    pos = txdec_ref.make_positions(toks, pad)
    assert pos.tolist() == [[8, 9, 10, 7, 7], [8, 9, 10, 11, 12]]
    d = 16
    half = d // 2
    full = torch.arange(20, dtype=torch.float).unsqueeze(1) * torch.exp(
        torch.arange(half, dtype=torch.float) * -(math.log(10000) / (half - 1))).unsqueeze(0)
    table = torch.cat([torch.sin(full), torch.cos(full)], dim=1)
    assert torch.equal(txdec_ref.sinusoidal_rows(pos, d), table[pos])


def test_layers_equal_torch_nn_transformer_decoder_layer():
    torch.manual_seed(0)
    vocab, d, ffn, n_layer, out_dim, pad, heads = 50, 32, 48, 2, 16, 49, 4
    w = txdec_ref.make_weights(vocab, d, ffn, n_layer, out_dim, pad, seed=3)
    toks = torch.randint(0, vocab - 1, (3, 7))
    toks[0, 5:] = pad
    enc = torch.randn(2, 3, d)
    logits = txdec_ref.forward(w, toks, enc, pad, heads, n_layer)
    # independent path: embeddings as above, then nn.TransformerDecoderLayer (post-norm, relu, no dropout)
    pos = txdec_ref.make_positions(toks, pad)
    x = math.sqrt(d) * w["embed_tokens.weight"][toks] + txdec_ref.sinusoidal_rows(pos, d) * pos.ne(pad).unsqueeze(-1)
    for i in range(n_layer):
        q = f"layers.{i}."
        layer = torch.nn.TransformerDecoderLayer(d, heads, ffn, dropout=0.0, activation="relu", batch_first=True,
                                                 norm_first=False).eval()
        with torch.no_grad():
            for att, mod in (("self_attn", layer.self_attn), ("encoder_attn", layer.multihead_attn)):
                mod.in_proj_weight.copy_(torch.cat([w[q + f"{att}.{p}_proj.weight"] for p in "qkv"]))
                mod.in_proj_bias.copy_(torch.cat([w[q + f"{att}.{p}_proj.bias"] for p in "qkv"]))
                mod.out_proj.weight.copy_(w[q + f"{att}.out_proj.weight"])
                mod.out_proj.bias.copy_(w[q + f"{att}.out_proj.bias"])
            for mine, theirs in (("self_attn_layer_norm", layer.norm1), ("encoder_attn_layer_norm", layer.norm2),
                                 ("final_layer_norm", layer.norm3)):
                theirs.weight.copy_(w[q + mine + ".weight"])
                theirs.bias.copy_(w[q + mine + ".bias"])
            layer.linear1.weight.copy_(w[q + "fc1.weight"]); layer.linear1.bias.copy_(w[q + "fc1.bias"])
            layer.linear2.weight.copy_(w[q + "fc2.weight"]); layer.linear2.bias.copy_(w[q + "fc2.bias"])
            causal = torch.triu(torch.full((7, 7), float("-inf")), 1)
            x = layer(x, enc.transpose(0, 1), tgt_mask=causal, tgt_key_padding_mask=toks.eq(pad))
    want = (x @ w["project_out_dim.weight"].t()) @ w["output_projection.weight"].t()
    valid = toks.ne(pad)
    assert float((logits - want)[valid].abs().max()) < 2e-4 * float(want.abs().max())


def test_encoder_layers_equal_torch_nn_transformer_encoder_layer():
    torch.manual_seed(1)
    d, ffn, n_layer, heads, pad = 32, 48, 2, 4, 9
    w = txdec_ref.make_encoder_weights(d, ffn, n_layer, seed=4)
    emb = torch.randn(3, 5, d)
    src = emb[..., 0]
    out = txdec_ref.encoder_forward(w, emb, src, pad, heads, n_layer)
    pos = txdec_ref.make_positions(src, pad)
    assert pos.tolist() == [[pad + 1 + t for t in range(5)]] * 3  # float features never equal the pad id
    x = math.sqrt(d) * emb + txdec_ref.sinusoidal_rows(pos, d)
    for i in range(n_layer):
        q = f"layers.{i}."
        layer = torch.nn.TransformerEncoderLayer(d, heads, ffn, dropout=0.0, activation="relu", batch_first=True,
                                                 norm_first=False).eval()
        with torch.no_grad():
            layer.self_attn.in_proj_weight.copy_(torch.cat([w[q + f"self_attn.{p}_proj.weight"] for p in "qkv"]))
            layer.self_attn.in_proj_bias.copy_(torch.cat([w[q + f"self_attn.{p}_proj.bias"] for p in "qkv"]))
            layer.self_attn.out_proj.weight.copy_(w[q + "self_attn.out_proj.weight"])
            layer.self_attn.out_proj.bias.copy_(w[q + "self_attn.out_proj.bias"])
            layer.norm1.weight.copy_(w[q + "self_attn_layer_norm.weight"]); layer.norm1.bias.copy_(w[q + "self_attn_layer_norm.bias"])
            layer.norm2.weight.copy_(w[q + "final_layer_norm.weight"]); layer.norm2.bias.copy_(w[q + "final_layer_norm.bias"])
            layer.linear1.weight.copy_(w[q + "fc1.weight"]); layer.linear1.bias.copy_(w[q + "fc1.bias"])
            layer.linear2.weight.copy_(w[q + "fc2.weight"]); layer.linear2.bias.copy_(w[q + "fc2.bias"])
            x = layer(x)
    assert float((out.transpose(0, 1) - x).abs().max()) < 2e-5 * float(x.abs().max())
