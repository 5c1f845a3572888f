"""Generates tests/golden/gpt2_*.npz with the INSTALLED huggingface transformers GPT2LMHeadModel
(the third-party code `vidsitu_code/hf_gpt2_fseq.py:150,165-203` calls; the reference pins
transformers==3.3.1, this container has a newer release of the same published model) on weights
from oracle.gpt2_ref.make_weights(seed).  Run from the repo root: python tests/golden/gen_gpt2_golden.py
Cases: a tiny model stored with its full logits, and a gpt2-medium-shaped 2-layer slice
(d 1024, 16 heads) stored as logits of a few positions + checksums (weights are re-derived from the
seed by the tests)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import gpt2_ref  # noqa: E402
from transformers import GPT2Config, GPT2LMHeadModel  # noqa: E402

CASES = [
    # name, vocab, n_pos, d, n_layer, n_head, R, L, seed
    ("tiny", 97, 32, 64, 2, 4, 3, 9, 11),
    ("tiny_l3_pad", 131, 40, 96, 3, 6, 4, 17, 12),
    ("medium_slice", 50300, 64, 1024, 2, 16, 5, 12, 13),
]
OUT = os.path.dirname(os.path.abspath(__file__))
for name, vocab, n_pos, d, n_layer, n_head, R, L, seed in CASES:
    w = gpt2_ref.make_weights(vocab, n_pos, d, n_layer, seed)
    cfg = GPT2Config(vocab_size=vocab, n_positions=n_pos, n_embd=d, n_layer=n_layer, n_head=n_head,
                     bos_token_id=0, eos_token_id=0, resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0)
    m = GPT2LMHeadModel(cfg).eval()
    sd = {k: torch.from_numpy(v) for k, v in w.items()}
    sd["lm_head.weight"] = sd["transformer.wte.weight"]
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all("attn.bias" in k or "masked_bias" in k for k in missing), (missing, unexpected)
    pad = 0
    rs = np.random.RandomState(seed + 100)
    toks = rs.randint(1, vocab, size=(R, L)).astype(np.int64)
    mask = np.ones((R, L), dtype=np.int64)
    for r in range(1, R):  # right padding of different lengths (row 0 stays full)
        n = L - (r * 3) % (L - 2)
        mask[r, n:] = 0
        toks[r, n:] = pad
    tt, mm = torch.from_numpy(toks), torch.from_numpy(mask)
    if vocab < 1000:  # training golden: Simple_TxDec.forward's loss (mdl_sf_base.py:653-667) + grads
        m.zero_grad()
        out = m(input_ids=tt, attention_mask=mm)
        lg = out.logits
        loss = torch.nn.functional.cross_entropy(lg[:, :-1].reshape(-1, vocab), tt[:, 1:].reshape(-1),
                                                 ignore_index=pad)
        loss.backward()
        grads = {"grad." + k: v.grad.numpy().copy() for k, v in m.named_parameters()}
        grads["loss"] = np.float32(loss.item())
    else:
        grads = {}
    with torch.no_grad():
        out = m(input_ids=tt, attention_mask=mm)
    logits = out.logits.numpy().astype(np.float32)
    ours = gpt2_ref.forward(w, toks, mask, n_head)
    err = float(np.abs(ours - logits)[mask.astype(bool)].max())
    print(f"{name}: logits {logits.shape}, |oracle - HF| max over valid rows {err:.3e}, "
          f"max |logit| {np.abs(logits).max():.3f}")
    save = dict(tokens=toks, mask=mask, dims=np.array([vocab, n_pos, d, n_layer, n_head, seed]), pad=np.int64(pad))
    save.update(grads)
    if logits.size < 200000:
        save["logits"] = logits
    else:
        save["logits_first64"] = logits[:, :, :64]
        save["logits_rowsum"] = logits.astype(np.float64).sum(-1)
        save["logits_argmax"] = logits.argmax(-1)
        save["logits_max"] = logits.max(-1)
    np.savez_compressed(os.path.join(OUT, f"gpt2_{name}.npz"), **save)
