"""The SlowFast trunk oracle restatement vs the facts that pin it (parity otherwise
UNPINNED -- see oracle/slowfast_ref.py header): published MAC / parameter counts, the
reference's attribute names and output widths.  CPU only (meta tensors)."""
import torch
from torch import nn

from oracle.slowfast_ref import (VideoTrunk, SFBaseRef, count_conv_macs_params, default_sf_cfg,
                                 slow_index)


def _meta_trunk(cfg):
    with torch.device("meta"):
        return VideoTrunk(cfg)


def test_slowfast_r50_macs_and_params():
    m = _meta_trunk(default_sf_cfg())
    fast = torch.empty(1, 3, 32, 224, 224, device="meta")
    macs, params = count_conv_macs_params(m, [fast[:, :, :8], fast])
    assert macs == 50307661824  # = 65.7 GMAC @256^2 x (224/256)^2 (SlowFast paper, 8x8 R50)
    assert params == 33583800
    assert sum(1 for x in m.modules() if isinstance(x, nn.Conv3d)) == 110
    assert sum(1 for x in m.modules() if isinstance(x, nn.BatchNorm3d)) == 110
    bn_affine = sum(p.numel() for n, p in m.named_parameters() if "bn" in n)
    assert bn_affine == 60688


def test_output_shapes_and_attribute_names():
    m = _meta_trunk(default_sf_cfg())
    fast = torch.empty(2, 3, 32, 224, 224, device="meta")
    out = m.forward_features([fast[:, :, :8], fast])
    assert [tuple(o.shape) for o in out] == [(2, 2048, 8, 7, 7), (2, 256, 32, 7, 7)]
    for name in "s1 s1_fuse s2 s2_fuse s3 s3_fuse s4 s4_fuse s5 pathway0_pool pathway1_pool".split():
        assert hasattr(m, name), name  # mdl_sf_base.py:22-33
    assert m.num_pathways == 2 and m.enable_detection is False
    keys = set(m.state_dict().keys())
    for k in ["s1.pathway0_stem.conv.weight", "s1.pathway1_stem.bn.running_var",
              "s1_fuse.conv_f2s.weight", "s1_fuse.bn.weight",
              "s2.pathway0_res0.branch1.weight", "s2.pathway0_res0.branch1_bn.bias",
              "s2.pathway0_res0.branch2.a.weight", "s2.pathway1_res2.branch2.c_bn.num_batches_tracked",
              "s5.pathway0_res2.branch2.b.weight"]:
        assert k in keys, k
    assert "s2.pathway0_res1.branch1.weight" not in keys  # shortcut conv only when shape changes


def test_i3d_and_tiny():
    m = _meta_trunk(default_sf_cfg("i3d", 50, 64, 8))
    x = torch.empty(1, 3, 8, 224, 224, device="meta")
    assert tuple(m.forward_features([x])[0].shape) == (1, 2048, 4, 7, 7)  # pathway0_pool [2,1,1]
    m = VideoTrunk(default_sf_cfg("i3d", "tiny", 8, 8))
    y = m.forward_features([torch.randn(2, 3, 8, 32, 32)])[0]
    assert tuple(y.shape) == (2, 256, 4, 4, 4)


def test_zero_init_final_bn_and_slow_index():
    m = VideoTrunk(default_sf_cfg("i3d", "tiny", 8, 8))
    assert float(m.s2.pathway0_res0.branch2.c_bn.weight.abs().sum()) == 0.0
    assert float(m.s2.pathway0_res0.branch2.a_bn.weight.sum()) == 8.0
    assert slow_index(32, 4).tolist() == [0, 4, 8, 13, 17, 22, 26, 31]  # video_utils.py:59-65


def test_sfbase_ref_logits_shape():
    mdl = SFBaseRef(default_sf_cfg("i3d", "tiny", 8, 8), n_vocab=17).eval()
    with torch.no_grad():
        out = mdl([torch.randn(2, 3, 8, 32, 32)])
    assert tuple(out.shape) == (2, 17)
