"""Verb-prediction evaluation step: softmax -> descending sort -> top-5.

Mirrors `EvalB.forward_one_batch` (`vidsitu_code/evl_vsitu.py:39-75`): this is the
output BASELINE calls "bit-exact event/verb indices".  The softmax + top-k runs in
one HIP kernel (`vs_softmax_topk`; ties resolve to the lowest index); the
per-rank pickle merge and text metrics of the reference are out of scope.
"""
import torch
from torch import nn

from . import ops


class EvalB(nn.Module):
    def __init__(self, cfg, comm, device=None):
        super().__init__()
        self.cfg = cfg
        self.full_cfg = cfg
        self.comm = comm
        self.device = device
        self.met_keys = ["Per_Ev_Top_1", "Per_Ev_Top_5"]
        self.topk_save = 5

    @torch.no_grad()
    def forward_one_batch(self, mdl, inp):
        mdl_out = mdl(inp)["mdl_out"]  # [B, E, V]
        B, E, V = mdl_out.shape
        probs, idx = ops.softmax_topk(mdl_out.reshape(B * E, V).float(), self.topk_save)
        probs = probs.view(B, E, -1).tolist()
        idx = idx.view(B, E, -1).tolist()
        symbols = getattr(self.comm.vb_id_vocab, "symbols", self.comm.vb_id_vocab)
        out = []
        for b, ann_idx in enumerate(inp["vseg_idx"].tolist()):
            out.append({
                "pred_vbs_ev": [[symbols[i] for i in ev] for ev in idx[b]],
                "pred_scores_ev": probs[b],
                "pred_ixs_ev": idx[b],
                "ann_idx": ann_idx,
            })
        return out

    @torch.no_grad()
    def forward(self, model, loss_fn, dl, dl_name="valid", rank=0, pred_path=None, mb=None):
        """Top-1 / top-5 per-event accuracy + mean loss over `dl` (a list of batches)."""
        model.eval()
        n, c1, c5, loss_sum, nb = 0, 0, 0, 0.0, 0
        for batch in dl:
            out = model(batch)
            loss_sum += float(loss_fn(out, batch)["loss"])
            nb += 1
            B, E, V = out["mdl_out"].shape
            _, idx = ops.softmax_topk(out["mdl_out"].reshape(B * E, V).float(), self.topk_save)
            lab = batch["label_tensor"].reshape(-1, 1)
            hit = idx == lab
            n += lab.numel()
            c1 += int(hit[:, 0].sum())
            c5 += int(hit.any(dim=1).sum())
        return ({"loss": loss_sum / max(nb, 1)},
                {"Per_Ev_Top_1": c1 / max(n, 1), "Per_Ev_Top_5": c5 / max(n, 1)})


class EvalB_Gen(EvalB):
    """`EvalB_Gen.forward_one_batch` (`vidsitu_code/evl_vsitu.py:159-214`): build a `SeqGenCustom`
    from `cfg.gen`, generate one SRL token sequence per event and decode it with the tokenizer.
    The SRL string parsing / caption metrics (cider, rouge, lea: external java + python packages)
    are out of scope; the integer token sequences -- the bit-exact part -- are returned."""

    def __init__(self, cfg, comm, device=None):
        super().__init__(cfg, comm, device)
        self.met_keys = ["cider", "rouge", "lea", "MacroVb_cider", "MacroArg_cider"]
        self.compute_loss = False

    @torch.no_grad()
    def forward(self, model, loss_fn, dl, dl_name="valid", rank=0, pred_path=None, mb=None):
        """The generation rows' validation pass (`evl_vsitu.py:77-145` as `EvalB_Gen` runs it): one
        beam search per batch; the teacher-forced LM loss of the same batches is reported beside it
        (these models return `{"loss", "logits"}`, not `mdl_out`).  The caption metrics named in
        `met_keys` need the external scorers and are out of scope: what comes back are the counts of
        the generated token sequences."""
        model.eval()
        loss_sum, nb, n_seq, n_tok = 0.0, 0, 0, 0
        for batch in dl:
            loss_sum += float(loss_fn(model(batch), batch)["loss"])
            nb += 1
            for rec in self.forward_one_batch(model, batch):
                for ev in rec["vb_output"].values():
                    n_seq += 1
                    n_tok += len(ev["tokens"])
        return ({"loss": loss_sum / max(nb, 1)},
                {"generated_sequences": n_seq, "tokens_per_sequence": n_tok / max(n_seq, 1)})

    @torch.no_grad()
    def forward_one_batch(self, mdl, inp):
        from .seq_gen import SeqGenCustom

        gen_kw = {k: self.cfg.gen[k] for k in self.cfg.gen}
        seq_gen = SeqGenCustom([mdl], tgt_dict=self.comm.gpt2_hf_tok, **gen_kw)
        out_sents = mdl.forward_gen(inp, seq_gen)  # [B, num_ev, 1, L] token ids, pad-filled
        wvoc = self.comm.gpt2_hf_tok
        out = []
        for pred_sent, ann_idx in zip(out_sents.tolist(), inp["vseg_idx"].tolist()):
            vb_output = {}
            for ev_ix, ev_sent in enumerate(pred_sent):
                assert len(ev_sent) == 1
                vb_output[f"Ev{ev_ix + 1}"] = {"tokens": ev_sent[0],
                                              "text": wvoc.decode(ev_sent[0], skip_special_tokens=True)}
            out.append({"ann_idx": ann_idx, "vb_output": vb_output})
        return out
