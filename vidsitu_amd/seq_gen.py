"""Host-side mirror of the reference's beam search, `SeqGenCustom` / `EnsembleModel`
(`vidsitu_code/seq_gen.py:29-544,775-913`; the search step itself is fairseq's
`BeamSearch.step`), same constructor arguments and output format
(`finalized[sent] = [{"tokens", "score", "attention", "alignment", "positional_scores"}, ...]`,
best first).  Differences, all deliberate:

* per-step scoring (temperature, log-softmax, pad / unk / max-len / min-len / prefix rules,
  cumulative scores, top 2*beam per row) is ONE HIP kernel (`vs_beam_topk`) instead of a chain
  of full-vocabulary torch ops; only the [bsz, beam * 2beam] candidate lists reach torch;
* the decoder keeps a KV cache (`use_kv_cache=True`, default) instead of re-encoding the prefix
  every step -- the reference's incremental states are empty dicts, hence falsy
  (`seq_gen.py:197-203`, `hf_gpt2_fseq.py:179-182`); `use_kv_cache=False` reproduces that;
* ties in the top-k (order unspecified in torch) go to the lowest flattened (beam, token) index;
* a single model (the reference never passes more than one), no n-gram blocking, no
  `match_source_len`, no constraints / lm_model (unused by the reference's configs: they raise).
"""
import math
import os

import torch
from torch import nn

from . import ops
from .hf_gpt2_fseq import KVCacheState


class EnsembleModel(nn.Module):
    """seq_gen.py:775-913 for one model."""

    def __init__(self, models):
        super().__init__()
        if len(models) != 1:
            raise NotImplementedError("ensembles are not on the VidSitu hot path")
        self.models_size = 1
        self.single_model = models[0]
        self.models = nn.ModuleList(models)

    def has_encoder(self):
        return self.single_model.use_encoder

    def max_decoder_positions(self):
        return self.single_model.max_decoder_positions()

    def forward_encoder(self, net_input):
        if not self.has_encoder():
            return None
        return [self.single_model.forward_encoder(net_input)]

    def reorder_encoder_out(self, encoder_outs, new_order):
        if not self.has_encoder():
            return []
        m = self.single_model
        fn = m.reorder_encoder_out if hasattr(m, "reorder_encoder_out") else m.encoder.reorder_encoder_out
        return [fn(encoder_outs[0], new_order)]

    def decoder_logits(self, tokens, encoder_outs, incremental_state):
        """Last-position logits [rows, V] (`forward_decoder` :812-853 before the log-softmax, which
        is fused into vs_beam_topk together with the temperature)."""
        enc = encoder_outs[0] if (self.has_encoder() and encoder_outs) else None
        out = self.single_model.decoder.forward(tokens, encoder_out=enc,
                                                incremental_state=incremental_state)
        return out[0][:, -1, :]


class _DeviceSearchSession:
    """Static buffers of one generation shape (batch, beam, lengths, vocabulary rules) and, when the
    decoder keeps a KV cache, one hipGraph per step: every kernel argument of a step (position,
    ping-pong parity, flags) is then a constant, and a replayed step costs one launch instead of
    ~230.  Use 1 of a shape runs eagerly (library / allocator / weight-copy warm-up), use 2 captures
    each step before replaying it, later uses only replay.  Graphs are dropped when the decoder's
    weight copies change (`_wt_epoch`)."""

    def __init__(self, key, dev):
        (self.bsz, self.beam, self.max_len, self.min_len, self.plen, self.V, self.pad, self.eos, self.unk,
         self.unk_penalty, self.temperature, self.normalize, self.len_penalty, self.kv, _) = key
        bsz, beam, max_len = self.bsz, self.beam, self.max_len
        rows, Lt, Ls = bsz * beam, max_len + 2, max_len + 1
        self.rows, self.Lt, self.Ls = rows, Lt, Ls
        i64, f32, i32, u8 = torch.long, torch.float32, torch.int32, torch.uint8
        self.tok = [torch.empty((rows, Lt), dtype=i64, device=dev) for _ in range(2)]
        self.sc = [torch.empty((rows, Ls), dtype=f32, device=dev) for _ in range(2)]
        self.anc = [torch.empty((rows, Lt), dtype=i32, device=dev) for _ in range(2)] if self.kv else None
        self.anc0 = torch.arange(rows, dtype=i32, device=dev).view(-1, 1).repeat(1, Lt) if self.kv else None
        self.ignore = torch.empty((bsz, beam), dtype=u8, device=dev)
        self.finished = torch.empty(bsz, dtype=u8, device=dev)
        self.nfin = torch.empty(bsz, dtype=i32, device=dev)
        self.remaining = torch.empty(1, dtype=i32, device=dev)
        self.fin_tok = torch.empty((bsz, beam, Ls), dtype=i64, device=dev)
        self.fin_score = torch.empty((bsz, beam), dtype=f32, device=dev)
        self.fin_pos = torch.empty((bsz, beam, Ls), dtype=f32, device=dev)
        self.fin_len = torch.empty((bsz, beam), dtype=i32, device=dev)
        self.reorder = torch.empty(rows, dtype=i64, device=dev)
        self.prefix = torch.empty((bsz, self.plen), dtype=i64, device=dev) if self.plen else None
        self.k = min(2 * beam, beam * self.V - 1, self.V - 1)
        self.state = None
        if self.kv:
            self.state = KVCacheState()
            self.state.max_len = Lt
        self.uses = 0
        self.graphs = {}
        self.pool = None
        self.epoch = None
        self.enc = None
        self.use_graphs = self.kv and os.environ.get("VS_GEN_GRAPHS", "1") != "0"

    def _reset(self, prefix_tokens, bos_token):
        self.tok[0].fill_(self.pad)
        self.tok[0][:, 0] = self.eos if bos_token is None else bos_token
        self.sc[0].zero_()
        for t in (self.ignore, self.finished, self.nfin, self.fin_score, self.fin_pos, self.fin_len):
            t.zero_()
        self.fin_tok.fill_(self.pad)
        self.remaining.fill_(self.bsz)
        if self.anc is not None:
            self.anc[0].copy_(self.anc0)
        if self.prefix is not None:
            self.prefix.copy_(prefix_tokens)

    def _step(self, gen, step):
        cur = step & 1
        st = self.state
        prev = self.tok[cur][:, : step + 1]
        if st is not None:
            st.len = step
            st.anc = self.anc[cur]
        else:
            prev = prev.contiguous()
        logits = gen.model.decoder_logits(prev, self.enc, st)
        forced, ban_eos = None, False
        if self.prefix is not None and step < self.plen and step < self.max_len:
            forced = self.prefix[:, step].unsqueeze(-1).repeat(1, self.beam).view(-1).contiguous()
        elif step < self.min_len:
            ban_eos = True
        cum = None if step == 0 else self.sc[cur][:, step - 1].contiguous()
        row_val, row_idx = ops.beam_topk(logits, cum, forced, self.k, self.pad, self.eos, self.unk,
                                         self.unk_penalty, self.temperature,
                                         eos_only=step >= self.max_len, ban_eos=ban_eos)
        anc_in, anc_out = (self.anc[cur], self.anc[1 - cur]) if self.anc is not None else (None, None)
        ops.beam_step(row_val, row_idx, self.tok[cur], self.tok[1 - cur], self.sc[cur], self.sc[1 - cur],
                      self.ignore, self.finished, self.nfin, self.remaining, self.fin_tok, self.fin_score,
                      self.fin_pos, self.fin_len, self.reorder, self.bsz, self.beam, self.k, self.V, step,
                      self.max_len, self.eos, self.normalize, self.len_penalty, anc_in=anc_in,
                      anc_out=anc_out)

    def run(self, gen, encoder_outs, prefix_tokens, bos_token):
        dec_model = getattr(gen.model.single_model.decoder, "model", None)
        epoch = getattr(dec_model, "_wt_epoch", 0)
        if epoch != self.epoch:  # weights changed (or first use): graphs are stale, warm up again
            self.graphs, self.pool, self.uses, self.epoch = {}, None, 0, epoch
        # a decoder that reads encoder outputs would need them at a fixed address inside the graphs
        dec = gen.model.single_model.decoder
        graphs = self.use_graphs and self.uses >= 1 and (not encoder_outs or
                                                         getattr(dec, "uses_encoder_out", True) is False)
        self.enc = encoder_outs
        self._reset(prefix_tokens, bos_token)
        if self.state is not None and hasattr(dec, "begin_incremental"):
            # per-generation constants (encoder attention) into the state's static buffers, eagerly
            dec.begin_incremental(self.state, encoder_outs[0] if encoder_outs else None, self.rows, self.Lt)
        for step in range(self.max_len + 1):
            if graphs:
                g = self.graphs.get(step)
                if g is None:
                    g = torch.cuda.CUDAGraph()
                    if self.pool is None:
                        self.pool = torch.cuda.graph_pool_handle()
                    with torch.cuda.graph(g, pool=self.pool):
                        self._step(gen, step)
                    self.graphs[step] = g
                g.replay()
            else:
                self._step(gen, step)
            if (step % 8 == 7 or step == self.max_len) and int(self.remaining) == 0:
                break
        self.uses += 1
        self.enc = None
        dev = self.nfin.device
        n_h, f_tok, f_sc, f_pos, f_len = (t.cpu() for t in (self.nfin, self.fin_tok, self.fin_score,
                                                            self.fin_pos, self.fin_len))
        finalized = []
        for sent in range(self.bsz):
            hyps = []
            for h in range(int(n_h[sent])):
                n = int(f_len[sent, h])
                hyps.append({"tokens": f_tok[sent, h, :n].to(dev), "score": f_sc[sent, h].to(dev),
                             "attention": torch.empty(0), "alignment": torch.empty(0),
                             "positional_scores": f_pos[sent, h, :n].to(dev)})
            order = torch.sort(torch.tensor([float(h["score"]) for h in hyps]), descending=True,
                               stable=True)[1]
            finalized.append([hyps[int(i)] for i in order])
        return finalized


class SeqGenCustom(nn.Module):
    def __init__(self, models, tgt_dict, beam_size=1, max_len_a=0, max_len_b=200, min_len=1,
                 normalize_scores=True, len_penalty=1.0, unk_penalty=0.0, temperature=1.0,
                 match_source_len=False, no_repeat_ngram_size=0, search_strategy=None, eos=None,
                 symbols_to_strip_from_output=None, lm_model=None, lm_weight=1.0, use_kv_cache=True,
                 device_search=True):
        super().__init__()
        self.model = models if isinstance(models, EnsembleModel) else EnsembleModel(models)
        self.tgt_dict = tgt_dict
        self.pad, self.unk = tgt_dict.pad(), tgt_dict.unk()
        self.eos = tgt_dict.eos() if eos is None else eos
        self.vocab_size = len(tgt_dict)
        self.beam_size = min(beam_size, self.vocab_size - 1)  # pad is never selected
        self.max_len_a, self.max_len_b, self.min_len = max_len_a, max_len_b, min_len
        self.normalize_scores, self.len_penalty = normalize_scores, len_penalty
        self.unk_penalty, self.temperature = unk_penalty, temperature
        assert temperature > 0, "--temperature must be greater than 0"
        if match_source_len or no_repeat_ngram_size > 0 or search_strategy is not None or lm_model is not None:
            raise NotImplementedError("option unused by the reference's configs (configs/vsitu_cfg.yml:76-85)")
        self.use_kv_cache = use_kv_cache
        # device_search: the bookkeeping between two decoder calls runs in one kernel per step
        # (vs_beam_step) with a host sync every 8 steps only; False = the torch-op mirror of the
        # reference's host loop below (one or more syncs per step)
        self.device_search = device_search and self.beam_size <= 32
        self.model.eval()

    @torch.no_grad()
    def forward(self, sample, prefix_tokens=None, bos_token=None):
        return self._generate(sample, prefix_tokens, bos_token=bos_token)

    @torch.no_grad()
    def generate(self, models, sample, **kwargs):
        return self._generate(sample, **kwargs)

    def _generate(self, sample, prefix_tokens=None, constraints=None, bos_token=None):
        if constraints is not None:
            raise NotImplementedError
        if self.device_search:
            return self._generate_device(sample, prefix_tokens, bos_token)
        return self._generate_host(sample, prefix_tokens, bos_token)

    def _generate_device(self, sample, prefix_tokens=None, bos_token=None):
        """Same search, state on the GPU: per step one decoder call, one
        vs_beam_topk, one vs_beam_step; the KV cache is never copied (ancestry table); finished
        sentences stay in the batch as idle rows instead of being removed; the host looks at the
        `remaining` counter every 8 steps.  With a KV cache the step sequence of one generation shape
        is captured in per-step hipGraphs on its second use (`_DeviceSearchSession`)."""
        src_tokens = sample["src_tokens"]
        dev = src_tokens.device
        bsz, src_len = src_tokens.size()[:2]
        max_len = min(int(self.max_len_a * src_len + self.max_len_b),
                      self.model.max_decoder_positions() - 1)
        assert self.min_len <= max_len, "min_len cannot be larger than max_len, please adjust these!"
        if prefix_tokens is not None and bool((prefix_tokens == self.eos).any()):
            raise NotImplementedError("eos inside prefix_tokens (unused by the reference's callers)")
        enc_inp = {k: v for k, v in sample.items() if "prev_tok" not in k}
        encoder_outs = self.model.forward_encoder(enc_inp)
        new_order = torch.arange(bsz, device=dev).view(-1, 1).repeat(1, self.beam_size).view(-1)
        encoder_outs = self.model.reorder_encoder_out(encoder_outs, new_order)  # beams of a sentence
        # share their encoder output, so later beam reorders leave it unchanged
        plen = 0 if prefix_tokens is None else prefix_tokens.size(1)
        dec = self.model.single_model.decoder
        key = (bsz, self.beam_size, max_len, self.min_len, plen, self.vocab_size, self.pad, self.eos,
               self.unk, float(self.unk_penalty), float(self.temperature), bool(self.normalize_scores),
               float(self.len_penalty), bool(self.use_kv_cache), str(dev))
        cache = dec.__dict__.setdefault("_vs_search_sessions", {})
        ses = cache.pop(key, None)
        if ses is None:
            ses = _DeviceSearchSession(key, dev)
            while len(cache) >= 2:  # a session owns a KV cache and a graph pool
                cache.pop(next(iter(cache)))
        cache[key] = ses  # most recently used last
        return ses.run(self, encoder_outs, prefix_tokens, bos_token)

    def _generate_host(self, sample, prefix_tokens=None, bos_token=None):
        src_tokens = sample["src_tokens"]
        dev = src_tokens.device
        bsz, src_len = src_tokens.size()[:2]
        beam, V = self.beam_size, self.vocab_size
        max_len = min(int(self.max_len_a * src_len + self.max_len_b),
                      self.model.max_decoder_positions() - 1)
        assert self.min_len <= max_len, "min_len cannot be larger than max_len, please adjust these!"
        enc_inp = {k: v for k, v in sample.items() if "prev_tok" not in k}
        encoder_outs = self.model.forward_encoder(enc_inp)
        new_order = torch.arange(bsz, device=dev).view(-1, 1).repeat(1, beam).view(-1)
        encoder_outs = self.model.reorder_encoder_out(encoder_outs, new_order)

        scores = torch.zeros(bsz * beam, max_len + 1, dtype=torch.float32, device=dev)
        tokens = torch.full((bsz * beam, max_len + 2), self.pad, dtype=torch.long, device=dev)
        tokens[:, 0] = self.eos if bos_token is None else bos_token
        cands_to_ignore = torch.zeros(bsz, beam, dtype=torch.bool, device=dev)
        finalized = [[] for _ in range(bsz)]
        finished = [False] * bsz
        num_remaining_sent = bsz
        cand_size = 2 * beam
        k = min(cand_size, beam * V - 1, V - 1)
        bbsz_offsets = (torch.arange(0, bsz, device=dev) * beam).unsqueeze(1)
        cand_offsets = torch.arange(0, cand_size, device=dev)
        state = None
        if self.use_kv_cache:
            state = KVCacheState()
            state.max_len = max_len + 2
        reorder_state = None
        batch_idxs = None

        for step in range(max_len + 1):  # one extra step for the EOS marker
            if reorder_state is not None:
                if batch_idxs is not None:  # beam indices after sentences were removed
                    corr = batch_idxs - torch.arange(batch_idxs.numel(), device=dev)
                    reorder_state.view(-1, beam).add_(corr.unsqueeze(-1) * beam)
                if state is not None:
                    self.model.single_model.decoder.reorder_incremental_state(state, reorder_state)
                encoder_outs = self.model.reorder_encoder_out(encoder_outs, reorder_state)

            logits = self.model.decoder_logits(tokens[:, : step + 1], encoder_outs, state)

            forced = None
            ban_eos = False
            if prefix_tokens is not None and step < prefix_tokens.size(1) and step < max_len:
                forced = prefix_tokens[:, step].unsqueeze(-1).repeat(1, beam).view(-1).contiguous()
                if bool((forced == self.eos).any()):
                    raise NotImplementedError("eos inside prefix_tokens (unused by the reference's callers)")
            elif step < self.min_len:
                ban_eos = True  # minimum length constraint (does not apply with prefix tokens)
            cum = None if step == 0 else scores[:, step - 1].contiguous()
            row_val, row_idx = ops.beam_topk(logits, cum, forced, k, self.pad, self.eos, self.unk,
                                             self.unk_penalty, self.temperature,
                                             eos_only=step >= max_len, ban_eos=ban_eos)
            # fairseq BeamSearch.step on the per-row lists: step 0 uses the first beam only
            rv, ri = row_val.view(bsz, beam, k), row_idx.view(bsz, beam, k)
            if step == 0:
                cand_scores, cand_indices = rv[:, 0, :], ri[:, 0, :]
                cand_beams = torch.zeros_like(cand_indices)
            else:
                flat_v = rv.reshape(bsz, beam * k)
                order = torch.sort(flat_v, dim=1, descending=True, stable=True)[1][:, :k]
                cand_scores = flat_v.gather(1, order)
                cand_indices = ri.reshape(bsz, beam * k).gather(1, order)
                cand_beams = order // k
            if k < cand_size:  # tiny vocabularies only
                padn = cand_size - k
                cand_scores = torch.cat([cand_scores, cand_scores.new_full((bsz, padn), -math.inf)], 1)
                cand_indices = torch.cat([cand_indices, cand_indices.new_zeros((bsz, padn))], 1)
                cand_beams = torch.cat([cand_beams, cand_beams.new_zeros((bsz, padn))], 1)
            cand_bbsz_idx = cand_beams + bbsz_offsets

            eos_mask = cand_indices.eq(self.eos) & cand_scores.ne(-math.inf)
            eos_mask[:, :beam][cands_to_ignore] = False
            eos_bbsz_idx = torch.masked_select(cand_bbsz_idx[:, :beam], mask=eos_mask[:, :beam])
            finalized_sents = []
            if eos_bbsz_idx.numel() > 0:
                eos_scores = torch.masked_select(cand_scores[:, :beam], mask=eos_mask[:, :beam])
                finalized_sents = self.finalize_hypos(step, eos_bbsz_idx, eos_scores, tokens, scores,
                                                      finalized, finished, beam, max_len)
                num_remaining_sent -= len(finalized_sents)
            assert num_remaining_sent >= 0
            if num_remaining_sent == 0:
                break
            assert step < max_len

            if len(finalized_sents) > 0:  # drop finished sentences from the batch
                new_bsz = bsz - len(finalized_sents)
                batch_mask = torch.ones(bsz, dtype=torch.bool, device=dev)
                batch_mask[finalized_sents] = False
                batch_idxs = torch.arange(bsz, device=dev).masked_select(batch_mask)
                eos_mask = eos_mask[batch_idxs]
                cand_beams = cand_beams[batch_idxs]
                bbsz_offsets = bbsz_offsets[:new_bsz]
                cand_bbsz_idx = cand_beams + bbsz_offsets
                cand_scores = cand_scores[batch_idxs]
                cand_indices = cand_indices[batch_idxs]
                if prefix_tokens is not None:
                    prefix_tokens = prefix_tokens[batch_idxs]
                cands_to_ignore = cands_to_ignore[batch_idxs]
                scores = scores.view(bsz, -1)[batch_idxs].view(new_bsz * beam, -1)
                tokens = tokens.view(bsz, -1)[batch_idxs].view(new_bsz * beam, -1)
                bsz = new_bsz
            else:
                batch_idxs = None

            # eos candidates (and ignored ones) sort behind every live candidate
            eos_mask[:, :beam] = ~((~cands_to_ignore) & (~eos_mask[:, :beam]))
            active_mask = eos_mask.long() * cand_size + cand_offsets[: eos_mask.size(1)]
            new_cands_to_ignore, active_hypos = torch.sort(active_mask, dim=1, stable=True)
            new_cands_to_ignore, active_hypos = new_cands_to_ignore[:, :beam], active_hypos[:, :beam]
            cands_to_ignore = new_cands_to_ignore.ge(cand_size)[:, :beam]
            assert (~cands_to_ignore).any(dim=1).all()
            active_bbsz_idx = torch.gather(cand_bbsz_idx, dim=1, index=active_hypos).view(-1)
            tokens[:, : step + 1] = torch.index_select(tokens[:, : step + 1], dim=0, index=active_bbsz_idx)
            tokens.view(bsz, beam, -1)[:, :, step + 1] = torch.gather(cand_indices, dim=1, index=active_hypos)
            if step > 0:
                scores[:, :step] = torch.index_select(scores[:, :step], dim=0, index=active_bbsz_idx)
            scores.view(bsz, beam, -1)[:, :, step] = torch.gather(cand_scores, dim=1, index=active_hypos)
            reorder_state = active_bbsz_idx

        for sent in range(len(finalized)):  # best first
            sc = torch.tensor([float(h["score"]) for h in finalized[sent]])
            order = torch.sort(sc, descending=True, stable=True)[1]
            finalized[sent] = [finalized[sent][int(i)] for i in order]
        return finalized

    def finalize_hypos(self, step, bbsz_idx, eos_scores, tokens, scores, finalized, finished,
                       beam_size, max_len):
        """seq_gen.py:579-697: store the hypotheses that just produced eos (at most beam_size per
        sentence); returns the batch positions of the sentences that are now complete."""
        tokens_clone = tokens.index_select(0, bbsz_idx)[:, 1: step + 2].clone()
        tokens_clone[:, step] = self.eos
        pos_scores = scores.index_select(0, bbsz_idx)[:, : step + 1].clone()
        pos_scores[:, step] = eos_scores
        pos_scores[:, 1:] = pos_scores[:, 1:] - pos_scores[:, :-1]
        if self.normalize_scores:
            eos_scores = eos_scores / (step + 1) ** self.len_penalty
        cum_unfin, prev = [], 0
        for f in finished:
            if f:
                prev += 1
            else:
                cum_unfin.append(prev)
        seen = []
        idx_list, score_list = bbsz_idx.tolist(), eos_scores.tolist()
        for i, idx in enumerate(idx_list):
            unfin_idx = idx // beam_size
            sent = unfin_idx + cum_unfin[unfin_idx]
            if (sent, unfin_idx) not in seen:
                seen.append((sent, unfin_idx))
            if len(finalized[sent]) < beam_size:
                finalized[sent].append({"tokens": tokens_clone[i], "score": eos_scores[i],
                                        "attention": torch.empty(0), "alignment": torch.empty(0),
                                        "positional_scores": pos_scores[i]})
        newly_finished = []
        for sent, unfin_idx in seen:
            if not finished[sent] and self.is_finished(step, unfin_idx, max_len, len(finalized[sent]),
                                                       beam_size):
                finished[sent] = True
                newly_finished.append(unfin_idx)
        return newly_finished

    def is_finished(self, step, unfin_idx, max_len, finalized_sent_len, beam_size):
        assert finalized_sent_len <= beam_size
        return finalized_sent_len == beam_size or step == max_len
