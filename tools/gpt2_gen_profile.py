"""Generation only (device-side search), for rocprofv3 --kernel-trace --stats:
   rocprofv3 --kernel-trace --stats -d gpurun_out/gen -- python3 tools/gpt2_gen_profile.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidsitu_amd import synth_data
from vidsitu_amd.extended_config import get_cfg
from vidsitu_amd.mdl_selector import get_mdl_loss_eval

B, beam, max_len = 2, 5, 60
dev = torch.device("cuda:0")
cfg = get_cfg({"task_type": "vb_arg", "mdl.mdl_name": "sfpret_txe_txd_vbarg", "mdl.tx_dec_type": "gpt2",
               "gen.beam_size": beam, "gen.max_len_b": max_len, "gen.min_len": max_len - 1})
comm = synth_data.make_comm(cfg)
sel = get_mdl_loss_eval(cfg)
torch.manual_seed(0)
mdl = sel["mdl"](cfg=cfg, comm=comm).to(dev).eval()
batch = synth_data.synth_srl_batch(comm, bs=B, n_ev=5, seq_len=60, device=dev)
evl = sel["evl"](cfg, comm, dev)
for it in range(int(os.environ.get("GEN_ITERS", "3"))):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = evl.forward_one_batch(mdl, batch)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"gen: {dt*1e3:.1f} ms")
