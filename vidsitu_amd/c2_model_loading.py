"""Caffe2 -> PyTorch checkpoint-name conversion for the video trunk.

The reference initialises `sf_mdl` from the Kinetics-pretrained SlowFast / I3D model-zoo files, which are
Caffe2 pickles: `Learner.__init__` -> `load_checkpoint(ckpt, model=mdl.sf_mdl, data_parallel=False,
convert_from_caffe2=cfg.sf_mdl.TRAIN.CHECKPOINT_TYPE == "caffe2")` (`utils/trn_utils.py:358-375`) and the
feature extractor does the same for `is_cu` runs (`vidsitu_code/feat_extractor.py:154-161`).  Both call
into the un-vendored third-party package `slowfast` (`slowfast/utils/checkpoint.py` +
`slowfast/utils/c2_model_loading.py`; `.gitmodules:9-11`, no pinned commit).

PARITY UNPINNED: this file restates the PUBLISHED behaviour of those two upstream functions -- an ordered
table of regular-expression rewrites applied one after the other to every blob name, then a
shape-checked, non-strict load -- from the upstream project's public source.  The reference tree holds no
Caffe2 file, golden name list or test for it; `tests/test_c2_loading.py` pins the table against the naming
scheme of the trunk's own state dict (a synthetic Caffe2-style pickle built with hand-written inverse
names round-trips tensor for tensor).

File format: `pickle.load(f, encoding="latin1")` -> {"blobs": {name: ndarray}} ; blobs whose name
contains "momentum", "lr" or "model_iter" are solver state and skipped silently.
"""
import pickle
import re

import numpy as np
import torch

# (pattern, replacement) applied IN ORDER with re.sub; later rows see the output of earlier ones
_PAIRS = [
    # ---- non-local blocks: 'nonlocal_conv4_5_theta_w' -> 's4.pathway0_nonlocal5.conv_theta.weight'
    [r"^nonlocal_conv([0-9]+)_([0-9]+)_(.*)", r"s\1.pathway0_nonlocal\2_\3"],
    [r"^(.*)_nonlocal([0-9]+)_(theta)(.*)", r"\1_nonlocal\2.conv_\3\4"],
    [r"^(.*)_nonlocal([0-9]+)_(g)(.*)", r"\1_nonlocal\2.conv_\3\4"],
    [r"^(.*)_nonlocal([0-9]+)_(phi)(.*)", r"\1_nonlocal\2.conv_\3\4"],
    [r"^(.*)_nonlocal([0-9]+)_(out)(.*)", r"\1_nonlocal\2.conv_\3\4"],
    [r"^(.*)_nonlocal([0-9]+)_(bn)_(.*)", r"\1_nonlocal\2.\3.\4"],
    # ---- lateral connections: 't_pool1_subsample_bn_rm' -> 's1_fuse.bn.running_mean'
    [r"^t_pool1_subsample_bn_(.*)", r"s1_fuse.bn.\1"],
    [r"^t_pool1_subsample_(.*)", r"s1_fuse.conv_f2s.\1"],
    [r"^t_res([0-9]+)_([0-9]+)_branch2c_bn_subsample_bn_(.*)", r"s\1_fuse.bn.\3"],
    [r"^t_res([0-9]+)_([0-9]+)_branch2c_bn_subsample_(.*)", r"s\1_fuse.conv_f2s.\3"],
    # ---- slow pathway (pathway0): 'res4_4_branch2c_bn_b' -> 's4.pathway0_res4.branch2.c_bn_b'
    [r"^res([0-9]+)_([0-9]+)_branch([0-9]+)([a-z])_(.*)", r"s\1.pathway0_res\2.branch\3.\4_\5"],
    [r"^res_conv1_bn_(.*)", r"s1.pathway0_stem.bn.\1"],
    [r"^conv1_xy(.*)", r"s1.pathway0_stem.conv_xy\1"],
    [r"^conv1_(.*)", r"s1.pathway0_stem.conv.\1"],
    [r"^res([0-9]+)_([0-9]+)_branch([0-9]+)_(.*)", r"s\1.pathway0_res\2.branch\3_\4"],
    [r"^res_conv1_(.*)", r"s1.pathway0_stem.conv.\1"],
    # ---- fast pathway (pathway1): the same names with a 't_' prefix
    [r"^t_res([0-9]+)_([0-9]+)_branch([0-9]+)([a-z])_(.*)", r"s\1.pathway1_res\2.branch\3.\4_\5"],
    [r"^t_res_conv1_bn_(.*)", r"s1.pathway1_stem.bn.\1"],
    [r"^t_conv1_(.*)", r"s1.pathway1_stem.conv.\1"],
    [r"^t_res([0-9]+)_([0-9]+)_branch([0-9]+)_(.*)", r"s\1.pathway1_res\2.branch\3_\4"],
    [r"^t_res_conv1_(.*)", r"s1.pathway1_stem.conv.\1"],
    # ---- head
    [r"pred_(.*)", r"head.projection.\1"],
    [r"(.*)b_bn_fc(.*)", r"\1se.fc\2"],
    [r"conv_5(.*)", r"head.conv_5\1"],
    [r"lin_5(.*)", r"head.lin_5\1"],
    # ---- parameter suffixes ('.' matches the '_' or '.' in front of the suffix)
    [r"(.*)bn.b\Z", r"\1bn.bias"],
    [r"(.*)bn.s\Z", r"\1bn.weight"],
    [r"(.*)bn.rm\Z", r"\1bn.running_mean"],
    [r"(.*)bn.riv\Z", r"\1bn.running_var"],
    [r"(.*)[\._]b\Z", r"\1.bias"],
    [r"(.*)[\._]w\Z", r"\1.weight"],
]


def get_name_convert_func():
    """-> f(caffe2 blob name) = pytorch state-dict key (upstream `c2_model_loading.get_name_convert_func`)."""
    pairs = [(re.compile(p), r) for p, r in _PAIRS]

    def convert(name):
        for pat, rep in pairs:
            name = pat.sub(rep, name)
        return name

    return convert


_SOLVER_STATE = ("momentum", "lr", "model_iter")


def convert_caffe2_blobs(blobs, model_state):
    """{caffe2 name: ndarray} -> ({state-dict key: tensor}, report).  A converted key is kept when the
    model owns it and the shapes agree after trailing singleton dims are appended to the blob (Linear
    weights stored for 1x1x1 convs and vice versa); BN statistics are tiled when the model's vector is a
    whole multiple of the blob's (upstream's Sub-BN rule).  report: {"loaded", "shape_mismatch",
    "not_in_model", "skipped"} lists of names."""
    conv = get_name_convert_func()
    out = {}
    report = {"loaded": [], "shape_mismatch": [], "not_in_model": [], "skipped": []}
    for key, blob in blobs.items():
        if any(s in key for s in _SOLVER_STATE):
            report["skipped"].append(key)
            continue
        ck = conv(key)
        if ck not in model_state:
            report["not_in_model"].append((key, ck))
            continue
        blob = np.asarray(blob)
        want = tuple(model_state[ck].shape)
        shape = tuple(blob.shape)
        if len(shape) < len(want):
            shape = shape + (1,) * (len(want) - len(shape))
            blob = blob.reshape(shape)
        if len(want) == 1 and len(shape) == 1 and want[0] > shape[0] and want[0] % shape[0] == 0:
            blob = np.concatenate([blob] * (want[0] // shape[0]))
            shape = tuple(blob.shape)
        if shape != want:
            report["shape_mismatch"].append((key, ck, tuple(blob.shape), want))
            continue
        out[ck] = torch.tensor(blob).clone()
        report["loaded"].append((key, ck))
    return out, report


def load_caffe2_checkpoint(path, model):
    """`load_checkpoint(path, model=mdl.sf_mdl, data_parallel=False, convert_from_caffe2=True)`
    (`utils/trn_utils.py:367-372`): converts, loads non-strictly, returns the report with the model keys
    that received nothing under "not_loaded" (`num_batches_tracked` excluded, as upstream)."""
    with open(path, "rb") as f:
        ckpt = pickle.load(f, encoding="latin1")
    if not isinstance(ckpt, dict) or "blobs" not in ckpt:
        raise ValueError(f"{path}: not a caffe2 checkpoint pickle (no 'blobs' entry)")
    msd = model.state_dict()
    sd, report = convert_caffe2_blobs(ckpt["blobs"], msd)
    report["not_loaded"] = sorted(k for k in set(msd) - set(sd) if "num_batches_tracked" not in k)
    model.load_state_dict(sd, strict=False)
    return report
