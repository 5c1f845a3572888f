"""Where do the D2D copies (__amd_rocclr_copyBuffer) of one eager step come from?  torch.profiler with stacks:
prints every aten::copy_ / clone / contiguous / cat call site of one step, grouped by Python frame.
usage: python tools/find_copies.py [feat_fwd|sf_txenc_train]"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "feat_fwd"
    from vidsitu_amd import synth_data
    from vidsitu_amd.extended_config import get_cfg
    from vidsitu_amd.mdl_selector import get_mdl_loss_eval
    from vidsitu_amd.optim import ArenaAdam, ParamArena
    dev = torch.device("cuda", 0)
    train = workload == "sf_txenc_train"
    ov = {"mdl.mdl_name": "sf_base_txenc" if train else "sf_base"}
    if train:
        ov["tx_dec.encoder_layers"] = 6
    cfg = get_cfg(ov)
    comm = synth_data.make_comm(cfg)
    torch.manual_seed(0)
    sel = get_mdl_loss_eval(cfg)
    mdl = sel["mdl"](cfg=cfg, comm=comm).to(dev)
    batch = synth_data.synth_batch(cfg, comm, bs=2, n_ev=4, seed=1234, device=dev, dtype=torch.bfloat16)
    if train:
        from vidsitu_amd.train_step import TrainStep
        mdl.train()
        arena = ParamArena(mdl)
        opt = ArenaAdam(arena, lr=cfg.train.lr, betas=(0.9, 0.99))
        ts = TrainStep(mdl, sel["loss"](cfg, comm), arena, opt, batch, world=1, use_dist=False)
        step = ts.step
    else:
        mdl.eval()

        def step():
            with torch.no_grad():
                return mdl.head(mdl.forward_encoder(batch))
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    names = collections.Counter()
    sites = collections.Counter()
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CUDA or "cuda" in str(e.device_type).lower():
            if "copy" in e.name.lower() or "memcpy" in e.name.lower() or "memset" in e.name.lower() or "fill" in e.name.lower():
                names[e.name] += 1
            continue
        if e.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::cat", "aten::zero_", "aten::fill_",
                      "aten::zeros", "aten::_foreach_add_", "aten::add_", "aten::to", "aten::_to_copy"):
            st = [s for s in (e.stack or []) if "vidsitu_amd" in s or "bench" in s]
            sites[(e.name, tuple(st[:2]))] += 1
    print("device-side copy/fill events:", dict(names))
    for (n, st), c in sites.most_common(40):
        print(f"{c:4d} {n:18s} {' <- '.join(st)}")


if __name__ == "__main__":
    main()
