"""Checkpoint files of the reference's trainer (`utils/trn_utils.py:631-716`):
one `torch.save`d dict {"model_state_dict", "optimizer_state_dict", "scheduler_state_dict"?, "num_it",
"num_epoch", "cfgtxt", "best_met"}.  Model keys and shapes are the reference's (the HIP modules keep the
upstream parameter names); the optimizer state is `torch.optim.Adam`'s (`ArenaAdam.state_dict`), so files
written here load into the reference's `Learner` and the other way round, with or without the
`module.` prefix of a DistributedDataParallel wrapper (`:637-642`)."""
import json
import os

import torch


def _strip_module(sd):
    if sd and all(k.split(".")[0] == "module" for k in sd):
        return {k.split(".", 1)[1]: v for k, v in sd.items()}
    return sd


def save_model_dict(path, mdl, optimizer=None, num_it=0, num_epoch=0, best_met=None, cfg=None):
    """`Learner.save_model_dict` (`trn_utils.py:699-716`)."""
    ckpt = {"model_state_dict": {k: v.detach().cpu().contiguous() for k, v in mdl.state_dict().items()},
            "num_it": int(num_it), "num_epoch": int(num_epoch), "best_met": best_met,
            "cfgtxt": json.dumps(cfg if isinstance(cfg, (dict, type(None))) else str(cfg))}
    if optimizer is not None:
        sd = optimizer.state_dict()
        for st in sd["state"].values():
            for k, v in st.items():
                if torch.is_tensor(v):
                    st[k] = v.detach().cpu()
        ckpt["optimizer_state_dict"] = sd
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as f:
        torch.save(ckpt, f)
    return ckpt


def load_model_dict(path, mdl, optimizer=None, load_opt=False, strict=True, arena=None):
    """`Learner.load_model_dict` (`trn_utils.py:631-697`): a missing file means "start from scratch"
    (returns None); otherwise the model (and, with load_opt, the optimizer) is restored in place and
    the bookkeeping {"num_it", "num_epoch", "best_met"} returned.  `arena`: the ParamArena the
    parameters live in -- its bf16 / transposed kernel copies are refreshed after the load."""
    if not os.path.exists(path):
        return None
    with open(path, "rb") as f:
        ckpt = torch.load(f, map_location="cpu", weights_only=False)
    mdl.load_state_dict(_strip_module(ckpt["model_state_dict"]), strict=strict)
    if arena is None and optimizer is not None:
        arena = getattr(optimizer, "arena", None)
    if arena is not None:
        arena.refresh()
    if load_opt:
        if optimizer is None or "optimizer_state_dict" not in ckpt:
            raise ValueError("load_opt needs an optimizer and a checkpoint that holds its state")
        optimizer.load_state_dict(ckpt["optimizer_state_dict"])
    return {k: ckpt.get(k) for k in ("num_it", "num_epoch", "best_met")}
