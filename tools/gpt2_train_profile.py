"""GPT-2-medium fine-tuning steps only (teacher-forced LM loss over 10 x 60 tokens, manual backward, fused
Adam), for `rocprofv3 --kernel-trace -- python3 tools/gpt2_train_profile.py`  (DEC=txdec: the fairseq-style
decoder)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidsitu_amd import synth_data
from vidsitu_amd.extended_config import get_cfg
from vidsitu_amd.mdl_selector import get_mdl_loss_eval
from vidsitu_amd.optim import ArenaAdam, ParamArena

dev = torch.device("cuda:0")
cfg = get_cfg({"task_type": "vb_arg", "mdl.mdl_name": "sfpret_txe_txd_vbarg",
               "mdl.tx_dec_type": os.environ.get("DEC", "gpt2")})
comm = synth_data.make_comm(cfg)
sel = get_mdl_loss_eval(cfg)
torch.manual_seed(0)
mdl = sel["mdl"](cfg=cfg, comm=comm).to(dev).train()
batch = synth_data.synth_srl_batch(comm, bs=2, n_ev=5, seq_len=60, device=dev)
arena = ParamArena(mdl, adopt_conv=False)
opt = ArenaAdam(arena, lr=1e-5)
loss_fn = sel["loss"](cfg, comm)
for it in range(int(os.environ.get("STEPS", "4"))):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    opt.zero_grad()
    loss = loss_fn(mdl(batch), batch)["loss"]
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    print(f"step {it}: {(time.perf_counter() - t0) * 1e3:.1f} ms")
