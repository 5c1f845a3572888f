"""GPU frame resize (first half) through the C-ABI (`vs_resize_bicubic_u8`) against
the oracle, the committed Pillow outputs and Pillow itself: bit-exact uint8; and the model-level input
`frms_ev_raw_u8` (decoded frames at source size) against resizing on the host first."""
import os

import numpy as np
import pytest
import torch

from oracle import resize_ref

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "resize_u8.npz")


def test_resize_equals_pillow_golden(dev):
    from vidsitu_amd import ops

    z = np.load(GOLD)
    for n in [k[:-3] for k in z.files if k.endswith("_in")]:
        want = z[n + "_out"]
        got = ops.resize_bicubic_u8(torch.from_numpy(z[n + "_in"]).to(dev), want.shape[0], want.shape[1])
        assert np.array_equal(got.cpu().numpy(), want), n


@pytest.mark.parametrize("shape", [(3, 2, 360, 640, 224, 224), (5, 240, 320, 224, 224), (2, 224, 398, 224, 224),
                                   (2, 100, 224, 224, 224), (4, 33, 47, 64, 16), (1, 224, 224, 224, 224),
                                   (1, 1080, 1920, 224, 224)])
def test_resize_equals_oracle_and_pillow(shape, dev):
    from vidsitu_amd import ops

    *lead, h, w, oh, ow = shape
    rs = np.random.RandomState(sum(shape))
    x = rs.randint(0, 256, tuple(lead) + (h, w, 3)).astype(np.uint8)
    got = ops.resize_bicubic_u8(torch.from_numpy(x).to(dev), oh, ow).cpu().numpy()
    assert got.shape == tuple(lead) + (oh, ow, 3)
    flat_in, flat_out = x.reshape(-1, h, w, 3), got.reshape(-1, oh, ow, 3)
    for f in range(flat_in.shape[0]):
        assert np.array_equal(flat_out[f], resize_ref.resize_bicubic_u8(flat_in[f], oh, ow)), f
    try:
        from PIL import Image
    except ImportError:
        return
    assert np.array_equal(flat_out[0], np.array(Image.fromarray(flat_in[0]).resize((ow, oh))))


def test_raw_frames_input_is_bitwise_the_resized_uint8_input(dev):
    from vidsitu_amd import synth_data
    from vidsitu_amd.extended_config import get_cfg
    from vidsitu_amd.mdl_selector import get_mdl_loss_eval

    cfg = get_cfg({"mdl.sf_mdl_name": "slow_fast_mini", "synth.num_verbs": 23})
    crop = int(cfg.sf_mdl.DATA.TRAIN_CROP_SIZE)
    comm = synth_data.make_comm(cfg)
    torch.manual_seed(0)
    mdl = get_mdl_loss_eval(cfg)["mdl"](cfg=cfg, comm=comm).to(dev).eval()
    t = cfg.sf_mdl.DATA.NUM_FRAMES
    raw = torch.randint(0, 256, (1, 2, t, 45, 80, 3), generator=torch.Generator().manual_seed(1)).to(torch.uint8)
    host = np.stack([resize_ref.resize_bicubic_u8(f, crop, crop) for f in raw.reshape(-1, 45, 80, 3).numpy()])
    host = torch.from_numpy(host).view(1, 2, t, crop, crop, 3)
    common = {"vseg_idx": torch.arange(1, device=dev)}
    with torch.no_grad():
        a = mdl({"frms_ev_raw_u8": raw.to(dev), **common})["mdl_out"]
        b = mdl({"frms_ev_fast_u8": host.to(dev), **common})["mdl_out"]
    assert torch.equal(a, b)
