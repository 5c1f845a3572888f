"""ORACLE (test infrastructure): numpy restatement of the reference's beam search,
`SeqGenCustom._generate` / `_prefix_tokens` / `finalize_hypos` / `is_finished`
(`vidsitu_code/seq_gen.py:190-544,546-577,579-697,700-716`) with `EnsembleModel.forward_decoder`
(`:812-872`: last-position logits / temperature -> log_softmax) and the search step of the
third-party dependency **fairseq** (`fairseq.search.BeamSearch.step`; un-vendored submodule
`.gitmodules:1-3`, fork github.com/TheShadow29/fairseq, no pinned commit).  Published algorithm of
that step, restated: step 0 uses only the first beam of every sentence; later steps add the
cumulative score of each beam to its log-probs; the top `2*beam` of the flattened `[beam*V]`
scores give (score, token = idx % V, beam = idx // V).
PARITY UNPINNED: fairseq is not installed, `seq_gen.py` cannot be imported here and the reference
holds no test or golden vector for generation; anchored on the call sites above and checked
against exhaustive search (tests/test_oracle_beam.py).  Ties in top-k (unspecified in torch) are
broken towards the lowest flattened index, here and in the HIP path.
"""
import math

import numpy as np

NEG_INF = -np.inf


def log_softmax(x):
    x = x.astype(np.float32)
    m = x.max(-1, keepdims=True)
    return (x - m) - np.log(np.exp(x - m).sum(-1, keepdims=True))


def topk_lowest_index(v, k):
    """Top-k of each row, descending, ties -> lowest index (stable sort of -v)."""
    idx = np.argsort(-v, axis=-1, kind="stable")[:, :k]
    return np.take_along_axis(v, idx, -1), idx


def generate(step_logits, bsz, vocab, pad, eos, unk, beam_size=1, max_len_a=0, max_len_b=200,
             min_len=1, normalize_scores=True, len_penalty=1.0, unk_penalty=0.0, temperature=1.0,
             prefix_tokens=None, bos_token=None, src_len=1, max_decoder_positions=1024):
    """step_logits(tokens [rows, step+1] int64, sent_ids [rows]) -> logits [rows, V] of the last
    position.  `sent_ids` = original sentence index of each row (rows shrink as sentences finish).
    Returns finalized[sent] = list of dicts(tokens, score, positional_scores), best first."""
    beam_size = min(beam_size, vocab - 1)
    max_len = min(int(max_len_a * src_len + max_len_b), max_decoder_positions - 1)
    assert min_len <= max_len
    scores = np.zeros((bsz * beam_size, max_len + 1), dtype=np.float32)
    tokens = np.full((bsz * beam_size, max_len + 2), pad, dtype=np.int64)
    tokens[:, 0] = eos if bos_token is None else bos_token
    cands_to_ignore = np.zeros((bsz, beam_size), dtype=bool)
    finalized = [[] for _ in range(bsz)]
    finished = [False] * bsz
    num_remaining = bsz
    cand_size = 2 * beam_size
    bbsz_offsets = (np.arange(bsz) * beam_size)[:, None]
    cand_offsets = np.arange(cand_size)
    sent_of_row = np.repeat(np.arange(bsz), beam_size)
    if prefix_tokens is not None:
        prefix_tokens = np.asarray(prefix_tokens)

    for step in range(max_len + 1):
        logits = step_logits(tokens[:, : step + 1], sent_of_row)
        lprobs = log_softmax(np.asarray(logits, dtype=np.float32) / np.float32(temperature))
        lprobs[lprobs != lprobs] = NEG_INF
        lprobs[:, pad] = NEG_INF
        lprobs[:, unk] -= unk_penalty
        if step >= max_len:
            lprobs[:, :eos] = NEG_INF
            lprobs[:, eos + 1:] = NEG_INF
        if prefix_tokens is not None and step < prefix_tokens.shape[1] and step < max_len:
            ptoks = np.repeat(prefix_tokens[:, step], beam_size)
            plp = lprobs[np.arange(len(ptoks)), ptoks]
            pm = ptoks != pad
            lprobs[pm] = NEG_INF
            lprobs[np.nonzero(pm)[0], ptoks[pm]] = plp[pm]
            em = ptoks == eos
            if em.any():  # prefix holds eos: make all beams of that sentence copies of beam 0
                emb = em.reshape(-1, beam_size)[:, 0]
                for arr in (tokens, scores, lprobs):
                    a3 = arr.reshape(-1, beam_size, arr.shape[-1])
                    a3[emb] = a3[emb][:, :1, :]
        elif step < min_len:
            lprobs[:, eos] = NEG_INF

        # fairseq BeamSearch.step
        lp3 = lprobs.reshape(bsz, beam_size, vocab)
        if step == 0:
            flat = lp3[:, 0, :].copy()
        else:
            flat = (lp3 + scores.reshape(bsz, beam_size, -1)[:, :, step - 1][:, :, None]).reshape(bsz, -1)
        k = min(cand_size, flat.shape[1] - 1)
        cand_scores, cidx = topk_lowest_index(flat, k)
        cand_beams, cand_indices = cidx // vocab, cidx % vocab
        cand_bbsz_idx = cand_beams + bbsz_offsets

        eos_mask = (cand_indices == eos) & (cand_scores != NEG_INF)
        eos_mask[:, :beam_size][cands_to_ignore] = False
        eos_bbsz_idx = cand_bbsz_idx[:, :beam_size][eos_mask[:, :beam_size]]
        finalized_sents = []
        if eos_bbsz_idx.size > 0:
            eos_scores = cand_scores[:, :beam_size][eos_mask[:, :beam_size]].copy()
            # ---- finalize_hypos
            tokens_clone = tokens[eos_bbsz_idx][:, 1: step + 2].copy()
            tokens_clone[:, step] = eos
            pos_scores = scores[eos_bbsz_idx][:, : step + 1].copy()
            pos_scores[:, step] = eos_scores
            pos_scores[:, 1:] = pos_scores[:, 1:] - pos_scores[:, :-1]
            if normalize_scores:
                eos_scores = eos_scores / np.float32((step + 1) ** len_penalty)
            cum_unfin, prev = [], 0
            for f in finished:
                if f:
                    prev += 1
                else:
                    cum_unfin.append(prev)
            seen = []
            for i in range(eos_bbsz_idx.shape[0]):
                unfin_idx = int(eos_bbsz_idx[i]) // beam_size
                sent = unfin_idx + cum_unfin[unfin_idx]
                if (sent, unfin_idx) not in seen:
                    seen.append((sent, unfin_idx))
                if len(finalized[sent]) < beam_size:
                    finalized[sent].append({"tokens": tokens_clone[i], "score": float(eos_scores[i]),
                                            "positional_scores": pos_scores[i]})
            for sent, unfin_idx in seen:
                if not finished[sent] and (len(finalized[sent]) == beam_size or step == max_len):
                    finished[sent] = True
                    finalized_sents.append(unfin_idx)
            num_remaining -= len(finalized_sents)
        assert num_remaining >= 0
        if num_remaining == 0:
            break
        assert step < max_len

        if finalized_sents:
            new_bsz = bsz - len(finalized_sents)
            batch_mask = np.ones(bsz, dtype=bool)
            batch_mask[finalized_sents] = False
            batch_idxs = np.nonzero(batch_mask)[0]
            eos_mask = eos_mask[batch_idxs]
            cand_beams = cand_beams[batch_idxs]
            bbsz_offsets = bbsz_offsets[:new_bsz]
            cand_bbsz_idx = cand_beams + bbsz_offsets
            cand_scores = cand_scores[batch_idxs]
            cand_indices = cand_indices[batch_idxs]
            if prefix_tokens is not None:
                prefix_tokens = prefix_tokens[batch_idxs]
            cands_to_ignore = cands_to_ignore[batch_idxs]
            scores = scores.reshape(bsz, -1)[batch_idxs].reshape(new_bsz * beam_size, -1)
            tokens = tokens.reshape(bsz, -1)[batch_idxs].reshape(new_bsz * beam_size, -1)
            sent_of_row = sent_of_row.reshape(bsz, -1)[batch_idxs].reshape(-1)
            bsz = new_bsz

        eos_mask[:, :beam_size] = ~((~cands_to_ignore) & (~eos_mask[:, :beam_size]))
        active_mask = eos_mask.astype(np.int64) * cand_size + cand_offsets[: eos_mask.shape[1]]
        order = np.argsort(active_mask, axis=1, kind="stable")[:, :beam_size]
        new_ignore = np.take_along_axis(active_mask, order, 1)
        active_hypos = order
        cands_to_ignore = (new_ignore >= cand_size)[:, :beam_size]
        assert (~cands_to_ignore).any(axis=1).all()
        active_bbsz_idx = np.take_along_axis(cand_bbsz_idx, active_hypos, 1).reshape(-1)
        tokens[:, : step + 1] = tokens[active_bbsz_idx][:, : step + 1]
        tokens.reshape(bsz, beam_size, -1)[:, :, step + 1] = np.take_along_axis(cand_indices, active_hypos, 1)
        if step > 0:
            scores[:, :step] = scores[active_bbsz_idx][:, :step]
        scores.reshape(bsz, beam_size, -1)[:, :, step] = np.take_along_axis(cand_scores, active_hypos, 1)

    for sent in range(len(finalized)):
        sc = np.array([h["score"] for h in finalized[sent]], dtype=np.float32)
        order = np.argsort(-sc, kind="stable")
        finalized[sent] = [finalized[sent][i] for i in order]
    return finalized
