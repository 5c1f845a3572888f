"""Generates tests/golden/greedy_gpt2_tiny.npz: huggingface `GPT2LMHeadModel.generate` (greedy, the call of
`vidsitu_code/mdl_sf_base.py:494-503 / 577-585`) of the INSTALLED transformers on weights from
oracle.gpt2_ref.make_weights(seed).  Random weights rarely emit a chosen eos by themselves, so the eos id of
each case is a token its own unconstrained continuation produces at a known step: some rows finish early
and are padded, others run to max_length; in the last case every row finishes and the output is shorter.  Run from the repo root: python tests/golden/gen_gpt2_greedy_golden.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import gpt2_ref  # noqa: E402
from transformers import GPT2Config, GPT2LMHeadModel  # noqa: E402

# vocab, n_pos, d, n_layer, n_head, rows, max_length, seed, (row, step) whose free-running token becomes eos
CASES = [(97, 32, 64, 2, 4, 6, 20, 11, (0, 4)), (131, 40, 96, 3, 6, 20, 31, 12, (3, 9)),
         (131, 40, 96, 3, 6, 5, 12, 12, None), (97, 32, 64, 2, 4, 4, 30, 11, "all")]
save = {"n_cases": np.int64(len(CASES))}
for ci, (vocab, n_pos, d, n_layer, n_head, rows, max_length, seed, pick) in enumerate(CASES):
    w = gpt2_ref.make_weights(vocab, n_pos, d, n_layer, seed)
    pad = vocab - 1
    cfg = GPT2Config(vocab_size=vocab, n_positions=n_pos, n_embd=d, n_layer=n_layer, n_head=n_head,
                     bos_token_id=0, eos_token_id=0, resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0)
    m = GPT2LMHeadModel(cfg).eval()
    sd = {k: torch.from_numpy(v) for k, v in w.items()}
    sd["lm_head.weight"] = sd["transformer.wte.weight"]
    m.load_state_dict(sd, strict=False)
    for first_seed in range(seed + 7, seed + 207):
        first = torch.from_numpy(np.random.RandomState(first_seed).randint(1, vocab - 1, size=(rows, 1))
                                 .astype(np.int64))
        kw = dict(input_ids=first, attention_mask=torch.ones_like(first), max_length=max_length, use_cache=True,
                  num_beams=1, num_return_sequences=1, do_sample=False, pad_token_id=pad)
        with torch.no_grad():
            free = m.generate(eos_token_id=None, **kw).numpy()
        if pick != "all":
            eos = int(free[pick[0], pick[1]]) if pick is not None else vocab - 2
            break
        # a token every row emits early: the loop must stop before max_length (first tokens re-drawn until
        # the rows share one)
        common = [t for t in np.unique(free[:, 1:]) if all((free[r, 1:max_length - 5] == t).any() for r in range(rows))]
        if common:
            eos = int(common[0])
            break
    else:
        raise SystemExit("no first tokens whose continuations share a token")
    with torch.no_grad():
        out = m.generate(eos_token_id=eos, **kw).numpy()
    ours = gpt2_ref.greedy_generate(w, first.numpy(), max_length, pad, eos, n_head)
    print(f"case {ci}: out {out.shape}, rows finished early {(out == pad).any(1).sum()} of {rows}, "
          f"oracle equal: {ours.shape == out.shape and (ours == out).all()}")
    save[f"c{ci}_dims"] = np.array([vocab, n_pos, d, n_layer, n_head, seed, max_length, pad, eos])
    save[f"c{ci}_first"] = first.numpy()
    save[f"c{ci}_out"] = out.astype(np.int64)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "greedy_gpt2_tiny.npz"), **save)
