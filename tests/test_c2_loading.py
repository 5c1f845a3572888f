"""Caffe2 -> PyTorch name conversion (`vidsitu_amd/c2_model_loading.py`; reference call sites
`utils/trn_utils.py:358-375`, `vidsitu_code/feat_extractor.py:154-161`).  PARITY UNPINNED (the table
restates the un-vendored `slowfast` package; the reference holds no Caffe2 file): what is pinned here is
that a Caffe2-style pickle written with the model zoo's naming scheme -- the inverse names are spelled out
by hand below, not derived from the table -- lands tensor for tensor on the trunk's state dict."""
import pickle
import re

import numpy as np
import pytest
import torch

from oracle import slowfast_ref
from vidsitu_amd import c2_model_loading as c2
from vidsitu_amd.trunk import VideoTrunk

_BN = {"weight": "s", "bias": "b", "running_mean": "rm", "running_var": "riv"}


def _c2_name(key, nblocks):
    """Model-zoo blob name of a pytorch state-dict key of the trunk (hand-written inverse)."""
    m = re.match(r"s1\.pathway([01])_stem\.(conv|bn)\.(\w+)$", key)
    if m:
        t = "t_" if m.group(1) == "1" else ""
        return f"{t}conv1_w" if m.group(2) == "conv" else f"{t}res_conv1_bn_{_BN[m.group(3)]}"
    m = re.match(r"s([1-4])_fuse\.(conv_f2s|bn)\.(\w+)$", key)
    if m:
        k = int(m.group(1))
        base = "t_pool1_subsample" if k == 1 else f"t_res{k}_{nblocks[k] - 1}_branch2c_bn_subsample"
        return f"{base}_w" if m.group(2) == "conv_f2s" else f"{base}_bn_{_BN[m.group(3)]}"
    m = re.match(r"s([2-5])\.pathway([01])_res(\d+)\.branch1(_bn)?\.(\w+)$", key)
    if m:
        t = "t_" if m.group(2) == "1" else ""
        base = f"{t}res{m.group(1)}_{m.group(3)}_branch1"
        return f"{base}_bn_{_BN[m.group(5)]}" if m.group(4) else f"{base}_w"
    m = re.match(r"s([2-5])\.pathway([01])_res(\d+)\.branch2\.([abc])(_bn)?\.(\w+)$", key)
    if m:
        t = "t_" if m.group(2) == "1" else ""
        base = f"{t}res{m.group(1)}_{m.group(3)}_branch2{m.group(4)}"
        return f"{base}_bn_{_BN[m.group(6)]}" if m.group(5) else f"{base}_w"
    m = re.match(r"s([2-5])\.pathway0_nonlocal(\d+)\.(conv_(theta|phi|g|out)|bn)\.(\w+)$", key)
    if m:
        base = f"nonlocal_conv{m.group(1)}_{m.group(2)}"
        if m.group(3) == "bn":
            return f"{base}_bn_{_BN[m.group(5)]}"
        return f"{base}_{m.group(4)}_{'w' if m.group(5) == 'weight' else 'b'}"
    raise AssertionError(f"no caffe2 name for {key}")


def _fake_c2_file(ref, path, depth_key):
    nblocks = dict(zip(range(2, 6), slowfast_ref.STAGE_DEPTH[depth_key]))
    g = torch.Generator().manual_seed(0)
    blobs, want = {}, {}
    for k, v in ref.state_dict().items():
        if "num_batches_tracked" in k:
            continue
        t = torch.randn(v.shape, generator=g)
        want[k] = t
        name = _c2_name(k, nblocks)
        assert name not in blobs, name
        blobs[name] = t.numpy().copy()
        blobs[name + "_momentum"] = np.zeros(1, np.float32)  # solver state: must be skipped
    # the classification head the trunk never builds, and solver scalars
    blobs["pred_w"] = np.zeros((400, 16), np.float32)
    blobs["pred_b"] = np.zeros((400,), np.float32)
    blobs["lr"] = np.float32(0.1)
    blobs["model_iter"] = np.int64(5)
    with open(path, "wb") as f:
        pickle.dump({"blobs": blobs}, f, protocol=2)
    return want


@pytest.mark.parametrize("arch,depth,nl", [("slowfast", 50, False), ("i3d", 50, True), ("i3d", "tiny", False)])
def test_caffe2_pickle_round_trips_onto_the_trunk(arch, depth, nl, tmp_path):
    cfg = slowfast_ref.default_sf_cfg(arch, depth, 8 if arch == "i3d" else 64, 8 if arch == "i3d" else 32)
    if nl:
        cfg.NONLOCAL.LOCATION = [[[]], [[1, 3]], [[1, 3, 5]], [[]]]
        cfg.NONLOCAL.INSTANTIATION = "softmax"
    ref = slowfast_ref.VideoTrunk(cfg)
    path = tmp_path / "model.pkl"
    want = _fake_c2_file(ref, path, depth)
    ours = VideoTrunk(cfg)
    rep = c2.load_caffe2_checkpoint(str(path), ours)
    assert not rep["shape_mismatch"] and not rep["not_loaded"], (rep["shape_mismatch"][:3], rep["not_loaded"][:3])
    assert sorted(ck for _, ck in rep["not_in_model"]) == ["head.projection.bias", "head.projection.weight"]
    assert len(rep["loaded"]) == len(want)
    sd = ours.state_dict()
    for k, v in want.items():
        assert torch.equal(sd[k], v), k


def test_known_names():
    f = c2.get_name_convert_func()
    for a, b in [("conv1_w", "s1.pathway0_stem.conv.weight"),
                 ("res_conv1_bn_riv", "s1.pathway0_stem.bn.running_var"),
                 ("t_conv1_w", "s1.pathway1_stem.conv.weight"),
                 ("res4_5_branch2c_bn_b", "s4.pathway0_res5.branch2.c_bn.bias"),
                 ("t_res3_0_branch1_w", "s3.pathway1_res0.branch1.weight"),
                 ("t_res3_0_branch1_bn_rm", "s3.pathway1_res0.branch1_bn.running_mean"),
                 ("t_pool1_subsample_w", "s1_fuse.conv_f2s.weight"),
                 ("t_res4_5_branch2c_bn_subsample_bn_s", "s4_fuse.bn.weight"),
                 ("nonlocal_conv3_1_theta_b", "s3.pathway0_nonlocal1.conv_theta.bias"),
                 ("nonlocal_conv4_5_bn_riv", "s4.pathway0_nonlocal5.bn.running_var"),
                 ("pred_w", "head.projection.weight")]:
        assert f(a) == b, (a, f(a), b)


def test_shape_mismatch_is_reported_not_loaded(tmp_path):
    cfg = slowfast_ref.default_sf_cfg("i3d", "tiny", 8, 8)
    ours = VideoTrunk(cfg)
    before = ours.s1.pathway0_stem.conv.weight.detach().clone()
    with open(tmp_path / "bad.pkl", "wb") as f:
        pickle.dump({"blobs": {"conv1_w": np.zeros((3, 3), np.float32)}}, f)
    rep = c2.load_caffe2_checkpoint(str(tmp_path / "bad.pkl"), ours)
    assert len(rep["shape_mismatch"]) == 1 and not rep["loaded"]
    assert torch.equal(ours.s1.pathway0_stem.conv.weight, before)
    with open(tmp_path / "nope.pkl", "wb") as f:
        pickle.dump({"weights": {}}, f)
    with pytest.raises(ValueError):
        c2.load_caffe2_checkpoint(str(tmp_path / "nope.pkl"), ours)


def test_load_sf_pretrained_follows_the_config_keys(tmp_path):
    """`checkpoint.load_sf_pretrained` = the `mdl.load_sf_pretrained` branch of the reference's Learner
    (`utils/trn_utils.py:358-375`): path and type from `sf_mdl.TRAIN.*`, a Caffe2 pickle lands on `mdl.sf_mdl`,
    the upstream head is tolerated, a stray tensor or a missing one raises, a missing file raises."""
    from vidsitu_amd import checkpoint, synth_data
    from vidsitu_amd.extended_config import get_cfg
    from vidsitu_amd.mdl_selector import get_mdl_loss_eval

    cfg = get_cfg({"mdl.sf_mdl_name": "slow_fast_nl_r50_8x8", "synth.num_verbs": 11})
    assert cfg.sf_mdl.TRAIN.CHECKPOINT_TYPE == "caffe2" and cfg.sf_mdl.TRAIN.CHECKPOINT_FILE_PATH.endswith(".pkl")
    comm = synth_data.make_comm(cfg)
    mdl = get_mdl_loss_eval(cfg)["mdl"](cfg=cfg, comm=comm)
    with pytest.raises(FileNotFoundError):
        checkpoint.load_sf_pretrained(cfg, mdl, log=lambda m: None)
    ref = slowfast_ref.VideoTrunk(cfg.sf_mdl)
    path = tmp_path / "SLOWFAST_CU_8x8_R50.pkl"
    want = _fake_c2_file(ref, path, 50)
    cfg.sf_mdl.TRAIN.CHECKPOINT_FILE_PATH = str(path)
    msgs = []
    checkpoint.load_sf_pretrained(cfg, mdl, log=msgs.append)
    assert msgs and "caffe2" in msgs[0]
    sd = mdl.sf_mdl.state_dict()
    for k, v in want.items():
        assert torch.equal(sd[k], v), k
    # a blob the trunk has no place for
    with open(path, "rb") as f:
        blobs = pickle.load(f)["blobs"]
    blobs["res9_0_branch2a_w"] = np.zeros((4, 4, 1, 1, 1), np.float32)
    bad = tmp_path / "stray.pkl"
    with open(bad, "wb") as f:
        pickle.dump({"blobs": blobs}, f, protocol=2)
    cfg.sf_mdl.TRAIN.CHECKPOINT_FILE_PATH = str(bad)
    with pytest.raises(ValueError, match="no place"):
        checkpoint.load_sf_pretrained(cfg, mdl, log=lambda m: None)
    # a trunk tensor the file lacks
    del blobs["res9_0_branch2a_w"], blobs["res2_0_branch2a_w"]
    with open(bad, "wb") as f:
        pickle.dump({"blobs": blobs}, f, protocol=2)
    with pytest.raises(ValueError, match="does not provide"):
        checkpoint.load_sf_pretrained(cfg, mdl, log=lambda m: None)
    # the slowfast torch format
    tpath = tmp_path / "trunk.pyth"
    torch.save({"model_state": {**{k: v for k, v in ref.state_dict().items()},
                                "head.projection.weight": torch.zeros(400, 2304)}}, tpath)
    cfg.sf_mdl.TRAIN.CHECKPOINT_FILE_PATH, cfg.sf_mdl.TRAIN.CHECKPOINT_TYPE = str(tpath), "pytorch"
    checkpoint.load_sf_pretrained(cfg, mdl, log=lambda m: None)
    for k, v in ref.state_dict().items():
        if "num_batches_tracked" not in k:
            assert torch.equal(mdl.sf_mdl.state_dict()[k], v), k
