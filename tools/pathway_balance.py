"""Per fork/join region of the two-stream trunk: time of the slow pathway (caller's stream) vs the
fast pathway (side stream), eager, HIP events.  Shows which chain each stage waits for."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidsitu_amd import synth_data, trunk as T
from vidsitu_amd.extended_config import get_cfg
from vidsitu_amd.mdl_selector import get_mdl_loss_eval
from vidsitu_amd.optim import ArenaAdam, ParamArena
dev = torch.device("cuda:0")
cfg = get_cfg({"mdl.mdl_name": "sf_base_txenc", "tx_dec.encoder_layers": 6})
comm = synth_data.make_comm(cfg)
torch.manual_seed(0)
sel = get_mdl_loss_eval(cfg)
mdl = sel["mdl"](cfg=cfg, comm=comm).to(dev).train()
loss_fn = sel["loss"](cfg, comm)
batch = synth_data.synth_batch(cfg, comm, bs=2, n_ev=4, device=dev, dtype=torch.bfloat16)
arena = ParamArena(mdl); opt = ArenaAdam(arena, lr=1e-4)
def step():
    opt.zero_grad(); loss_fn(mdl(batch), batch)["loss"].backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
recs = []
orig_fork, orig_join = T._Fork.fork, T._Fork.join
def fork(self):
    orig_fork(self)
    if self.side is not None:
        self._e0 = torch.cuda.Event(enable_timing=True); self._e0.record(self.main)
def join(self, keep=None):
    if self.side is not None and self.active:
        em, es = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        T._WgradLanes.join_all()
        em.record(self.main); es.record(self.side)
        recs.append((self._e0, em, es))
    orig_join(self, keep)
T._Fork.fork, T._Fork.join = fork, join
# park the GPU so that launches are queued back to back
torch.cuda._sleep(int(3e8))
step()
torch.cuda.synchronize()
names = ["fwd stems", "fwd s2", "fwd s3", "fwd s4", "fwd s5", "bwd s5", "bwd s4", "bwd s3", "bwd s2", "bwd stems"]
ts = tf = tw = 0.0
for i, (e0, em, es) in enumerate(recs):
    a, b = e0.elapsed_time(em), e0.elapsed_time(es)
    ts += a; tf += b; tw += max(a, b)
    print(f"{names[i] if i < len(names) else i:10s} slow {a*1e3:8.1f} us   fast {b*1e3:8.1f} us   {'FAST is longer' if b > a else ''}")
print(f"sum slow {ts:.3f} ms, sum fast {tf:.3f} ms, sum of max {tw:.3f} ms")
