"""Checkpoint files of the reference's trainer (`utils/trn_utils.py:631-716`):
one `torch.save`d dict {"model_state_dict", "optimizer_state_dict", "scheduler_state_dict"?, "num_it",
"num_epoch", "cfgtxt", "best_met"}.  Model keys and shapes are the reference's (the HIP modules keep the
upstream parameter names); the optimizer state is `torch.optim.Adam`'s, indexed over the reference's
`mdl.parameters()` order (`ArenaAdam.state_dict`), so files written here load into the reference's
`Learner` and the other way round, with or without the `module.` prefix of a DistributedDataParallel
wrapper (`:637-642`).

One asymmetry is bridged here: the upstream SlowFast / ResNet constructor builds a classification head
`sf_mdl.head.projection.{weight,bias}` that `forward_features` never runs (`mdl_sf_base.py:21-34`), so every
reference checkpoint carries those two tensors and the reference's strict load expects them; `VideoTrunk`
does not build them.  `load_model_dict` drops them before the strict check, `save_model_dict` writes
zero placeholders of the upstream shape (`reference_only_params`)."""
import json
import os

import torch


def _strip_module(sd):
    if sd and all(k.split(".")[0] == "module" for k in sd):
        return {k.split(".", 1)[1]: v for k, v in sd.items()}
    return sd


def reference_only_keys(mdl):
    """{state_dict key: shape} of the tensors only the reference's model owns (see the module docstring)."""
    out = {}
    for mname, m in mdl.named_modules():
        fn = getattr(m, "reference_only_params", None)
        if callable(fn):
            for suffix, shape in fn():
                out[(mname + "." if mname else "") + suffix] = tuple(shape)
    return out


def save_model_dict(path, mdl, optimizer=None, num_it=0, num_epoch=0, best_met=None, cfg=None):
    """`Learner.save_model_dict` (`trn_utils.py:699-716`)."""
    msd = {k: v.detach().cpu().contiguous() for k, v in mdl.state_dict().items()}
    for k, shape in reference_only_keys(mdl).items():  # placeholders the reference's strict load expects
        msd.setdefault(k, torch.zeros(shape))
    ckpt = {"model_state_dict": msd,
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
    import pickle

    with open(path, "rb") as f:
        # tensors, python scalars / containers only: the reference's files hold nothing else -- except that its
        # `best_met` / scheduler bookkeeping may be a numpy scalar (`utils/trn_utils.py:699-716` saves whatever the
        # metric function returned).  Such a file is refused by the safe unpickler: say so, and how to load it.
        try:
            ckpt = torch.load(f, map_location="cpu", weights_only=True)
        except pickle.UnpicklingError as e:
            if os.environ.get("VS_CKPT_UNSAFE_LOAD") != "1":
                raise RuntimeError(
                    f"{path}: the checkpoint holds objects the safe (weights_only) loader refuses -- typically a numpy "
                    f"scalar in `best_met` written by the reference trainer.  If the file is trusted, set "
                    f"VS_CKPT_UNSAFE_LOAD=1 to load it with the full unpickler.  ({e})") from e
            f.seek(0)
            ckpt = torch.load(f, map_location="cpu", weights_only=False)
    msd = _strip_module(ckpt["model_state_dict"])
    ref_only = reference_only_keys(mdl)
    for k, shape in ref_only.items():
        if k in msd:
            if tuple(msd[k].shape) != shape:
                raise ValueError(f"checkpoint tensor {k}: shape {tuple(msd[k].shape)}, the upstream model has {shape}")
            msd = {kk: v for kk, v in msd.items() if kk != k}
    mdl.load_state_dict(msd, strict=strict)
    if arena is None and optimizer is not None:
        arena = getattr(optimizer, "arena", None)
    if arena is not None:
        arena.refresh()
    if load_opt:
        if optimizer is None or "optimizer_state_dict" not in ckpt:
            raise ValueError("load_opt needs an optimizer and a checkpoint that holds its state")
        optimizer.load_state_dict(ckpt["optimizer_state_dict"])
    return {k: ckpt.get(k) for k in ("num_it", "num_epoch", "best_met")}


def load_sf_pretrained(cfg, mdl, log=print):
    """`Learner.__init__`'s `elif self.cfg.mdl["load_sf_pretrained"]` branch for `task_type == "vb"`
    (`utils/trn_utils.py:358-375`; the feature extractor does the same at `feat_extractor.py:154-161`): the trunk
    `mdl.sf_mdl` starts from the Kinetics weights named by `cfg.sf_mdl.TRAIN.CHECKPOINT_FILE_PATH` -- a model-zoo Caffe2
    pickle (`CHECKPOINT_TYPE: caffe2`, names converted by `c2_model_loading`) or a slowfast-format torch file
    (`{"model_state": ...}`).  The upstream classification head (`head.projection.*`, which `forward_features` never
    runs and `VideoTrunk` does not build) is the only thing a file may hold beyond the trunk: anything else unknown
    raises, as upstream's assert does."""
    from .c2_model_loading import load_caffe2_checkpoint

    tr = cfg.sf_mdl.TRAIN
    path = tr.CHECKPOINT_FILE_PATH
    if not path or not os.path.exists(path):
        raise FileNotFoundError(f"mdl.load_sf_pretrained: sf_mdl.TRAIN.CHECKPOINT_FILE_PATH = {path!r} does not exist")
    trunk = mdl.sf_mdl
    if tr.CHECKPOINT_TYPE == "caffe2":
        report = load_caffe2_checkpoint(path, trunk)
        stray = [k for k, ck in report["not_in_model"] if not ck.startswith("head.projection")]
        missing = report["not_loaded"]
    else:
        with open(path, "rb") as f:
            ckpt = torch.load(f, map_location="cpu", weights_only=True)
        sd = _strip_module(ckpt.get("model_state", ckpt.get("model_state_dict", ckpt)))
        sd = {k: v for k, v in sd.items() if not k.startswith("head.projection")}
        res = trunk.load_state_dict(sd, strict=False)
        stray, missing = list(res.unexpected_keys), [k for k in res.missing_keys if "num_batches_tracked" not in k]
    if stray:
        raise ValueError(f"{path}: tensors the trunk has no place for: {stray[:8]}{' ...' if len(stray) > 8 else ''}")
    if missing:
        raise ValueError(f"{path}: trunk parameters the file does not provide: {missing[:8]}{' ...' if len(missing) > 8 else ''}")
    if hasattr(trunk, "refresh_weights") and next(trunk.parameters()).is_cuda:
        trunk.refresh_weights()
    log(f"loaded pretrained trunk weights from {path} ({tr.CHECKPOINT_TYPE})")
