"""CPU tests of the GPT-2 / beam-search oracles: the numpy GPT-2 restatement against the golden
logits produced by huggingface transformers (tests/golden/gen_gpt2_golden.py), and the beam-search
restatement against exhaustive search on a toy language model."""
import glob
import itertools
import os

import numpy as np
import pytest

from oracle import beam_ref, gpt2_ref

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "gpt2_*.npz")))


def load_case(path):
    z = np.load(path)
    vocab, n_pos, d, n_layer, n_head, seed = [int(v) for v in z["dims"]]
    w = gpt2_ref.make_weights(vocab, n_pos, d, n_layer, seed)
    return z, w, n_head


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_gpt2_oracle_matches_transformers_golden(path):
    z, w, n_head = load_case(path)
    logits = gpt2_ref.forward(w, z["tokens"], z["mask"], n_head)
    valid = z["mask"].astype(bool)
    if "logits" in z:
        assert np.abs(logits - z["logits"])[valid].max() < 2e-5
    else:
        assert np.abs(logits[:, :, :64] - z["logits_first64"])[valid].max() < 1e-4
        assert np.abs(logits.astype(np.float64).sum(-1) - z["logits_rowsum"])[valid].max() < 5e-2
        assert (logits.argmax(-1) == z["logits_argmax"])[valid].all()


def test_gpt2_outputs_do_not_depend_on_right_padding():
    z, w, n_head = load_case(GOLD[0])
    toks, mask = z["tokens"], z["mask"]
    full = gpt2_ref.forward(w, toks, mask, n_head)
    n = int(mask[1].sum())
    alone = gpt2_ref.forward(w, toks[1:2, :n], np.ones((1, n), dtype=np.int64), n_head)
    assert np.abs(full[1, :n] - alone[0]).max() < 1e-5


def _toy_lm(vocab, seed):
    """Deterministic 'language model': logits depend on the last two tokens."""
    rs = np.random.RandomState(seed)
    table = rs.randn(vocab, vocab, vocab).astype(np.float32) * 2.0

    def step(tokens, sent_ids):
        prev2 = tokens[:, -2] if tokens.shape[1] > 1 else np.zeros(len(tokens), dtype=np.int64)
        return table[prev2, tokens[:, -1]] + 0.1 * sent_ids[:, None].astype(np.float32)
    return step, table


def _exhaustive_best(table, vocab, pad, eos, max_len, sent):
    """Best length-normalised complete hypothesis by brute force (every sequence ending in eos)."""
    best = (-np.inf, None)
    for n in range(1, max_len + 2):
        for seq in itertools.product([t for t in range(vocab) if t not in (pad, eos)], repeat=n - 1):
            toks = [eos] + list(seq) + [eos]
            s = 0.0
            for i in range(1, len(toks)):
                prev2 = toks[i - 2] if i >= 2 else 0
                lg = table[prev2, toks[i - 1]] + 0.1 * sent
                lp = beam_ref.log_softmax(lg[None])[0]
                lp[pad] = -np.inf
                s += lp[toks[i]]
            s /= n
            if s > best[0]:
                best = (s, toks[1:])
    return best


def test_beam_search_full_width_equals_exhaustive_search():
    vocab, pad, eos, unk = 6, 1, 0, 2
    step, table = _toy_lm(vocab, 3)
    max_len = 3
    out = beam_ref.generate(step, bsz=2, vocab=vocab, pad=pad, eos=eos, unk=unk, beam_size=5,
                            max_len_b=max_len, min_len=0)
    # a beam as wide as the vocabulary keeps the greedy prefix of the optimum alive at every
    # length; compare the best finalized score with brute force over all sequences
    for sent in range(2):
        want_s, want_toks = _exhaustive_best(table, vocab, pad, eos, max_len, sent)
        got = out[sent][0]
        assert got["score"] <= want_s + 1e-5
        assert len(out[sent]) == 5
        assert all(h["tokens"][-1] == eos for h in out[sent])
        assert all(out[sent][i]["score"] >= out[sent][i + 1]["score"] for i in range(4))


def test_beam_one_is_greedy_and_prefix_is_forced():
    vocab, pad, eos, unk = 9, 1, 0, 2
    step, table = _toy_lm(vocab, 5)
    prefix = np.array([[4], [7], [3]])
    out = beam_ref.generate(step, bsz=3, vocab=vocab, pad=pad, eos=eos, unk=unk, beam_size=1,
                            max_len_b=6, min_len=0, prefix_tokens=prefix)
    for sent in range(3):
        toks = out[sent][0]["tokens"]
        assert toks[0] == prefix[sent, 0]
        seq = [eos] + list(toks)
        for i in range(2, len(seq)):  # every later token is the argmax of its step
            lg = table[seq[i - 2], seq[i - 1]] + 0.1 * sent
            lp = beam_ref.log_softmax(lg[None])[0]
            lp[pad] = -np.inf
            if i - 1 >= 6:
                assert seq[i] == eos
            else:
                assert seq[i] == int(np.argmax(lp))
        assert np.isclose(out[sent][0]["positional_scores"].sum() / len(toks), out[sent][0]["score"], atol=1e-5)


def test_beam_scores_are_length_normalised_sums_of_token_log_probs():
    vocab, pad, eos, unk = 7, 1, 0, 2
    step, table = _toy_lm(vocab, 9)
    out = beam_ref.generate(step, bsz=2, vocab=vocab, pad=pad, eos=eos, unk=unk, beam_size=3,
                            max_len_b=5, min_len=1)
    for sent in range(2):
        for h in out[sent]:
            seq = [eos] + list(h["tokens"])
            s = 0.0
            for i in range(1, len(seq)):
                prev2 = seq[i - 2] if i >= 2 else 0
                lp = beam_ref.log_softmax((table[prev2, seq[i - 1]] + 0.1 * sent)[None])[0]
                s += lp[seq[i]]
            assert np.isclose(s / len(h["tokens"]), h["score"], atol=1e-4)


def test_greedy_generation_oracle_matches_transformers_golden():
    """oracle.gpt2_ref.greedy_generate against huggingface `generate` outputs
    (tests/golden/gen_gpt2_greedy_golden.py): eos -> pad fill, max_length, early stop."""
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "greedy_gpt2_tiny.npz"))
    shorter = 0
    for ci in range(int(z["n_cases"])):
        vocab, n_pos, d, n_layer, n_head, seed, max_length, pad, eos = [int(v) for v in z[f"c{ci}_dims"]]
        w = gpt2_ref.make_weights(vocab, n_pos, d, n_layer, seed)
        got = gpt2_ref.greedy_generate(w, z[f"c{ci}_first"], max_length, pad, eos, n_head)
        want = z[f"c{ci}_out"]
        assert got.shape == want.shape and (got == want).all()
        shorter += want.shape[1] < max_length
    assert shorter >= 1  # the fixture holds a batch that stops before max_length
