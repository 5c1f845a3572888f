"""Timing of beam-search generation through the plugin surface (BASELINE configs[4] shape on one GPU:
B videos x 5 events, beam 5, up to 60 tokens) with the GPT-2-medium decoder, or -- `DEC=txdec` in the
environment -- the fairseq-style 3-layer TransformerDecoder (the reference's default `tx_dec_type`).
Informational, not bench.py."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidsitu_amd import synth_data
from vidsitu_amd.extended_config import get_cfg
from vidsitu_amd.mdl_selector import get_mdl_loss_eval

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
beam = int(sys.argv[2]) if len(sys.argv) > 2 else 5
max_len = int(sys.argv[3]) if len(sys.argv) > 3 else 60
dev = torch.device("cuda:0")
cfg = get_cfg({"task_type": "vb_arg", "mdl.mdl_name": "sfpret_txe_txd_vbarg",
               "mdl.tx_dec_type": os.environ.get("DEC", "gpt2"),
               "gen.beam_size": beam, "gen.max_len_b": max_len, "gen.min_len": max_len - 1})
comm = synth_data.make_comm(cfg)
sel = get_mdl_loss_eval(cfg)
torch.manual_seed(0)
mdl = sel["mdl"](cfg=cfg, comm=comm).to(dev).eval()
batch = synth_data.synth_srl_batch(comm, bs=B, n_ev=5, seq_len=60, device=dev)
evl = sel["evl"](cfg, comm, dev)
outs = {}
for dsearch in (False, True):
    evl.cfg.gen.device_search = dsearch
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = evl.forward_one_batch(mdl, batch)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        ntok = sum(len(v["tokens"]) for r in out for v in r["vb_output"].values())
        print(f"gen ({'device' if dsearch else 'host'} search): {B} videos x 5 events, beam {beam}: "
              f"{dt*1e3:.1f} ms, {ntok} output tokens, {5*B*beam*max_len/dt:.0f} beam-token steps/s")
    outs[dsearch] = [v["tokens"] for r in out for v in r["vb_output"].values()]
print("device search tokens == host search tokens:", outs[True] == outs[False])
torch.cuda.synchronize(); t0 = time.perf_counter()
o = mdl(batch)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"teacher-forced loss forward: {5*B}x60 tokens in {dt*1e3:.1f} ms, loss {float(o['loss']):.4f}")
# one fine-tuning step of the language model (teacher-forced loss, manual backward, fused Adam)
from vidsitu_amd.optim import ArenaAdam, ParamArena
mdl.train()
arena = ParamArena(mdl, adopt_conv=False)
opt = ArenaAdam(arena, lr=1e-5)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    opt.zero_grad()
    loss = sel["loss"](cfg, comm)(mdl(batch), batch)["loss"]
    loss.backward()
    opt.step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"train step {it}: {5*B}x60 tokens fwd+bwd+Adam in {dt*1e3:.1f} ms, loss {float(loss.detach()):.4f}")
