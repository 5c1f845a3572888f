"""The frame-resize oracle (`oracle/resize_ref.py`, Pillow's bicubic resampling of 8-bit RGB as used by
`VsituDS.read_img`, dat_loader.py:183-191) against the committed Pillow outputs and, when Pillow is
importable, against Pillow itself.  The host coefficient tables of the C-ABI (`vs_resize_coeffs`, no GPU
needed) must equal the oracle's.  Bit-exact."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import resize_ref

GOLD = os.path.join(os.path.dirname(__file__), "golden", "resize_u8.npz")


def test_oracle_equals_pillow_golden():
    z = np.load(GOLD)
    names = [k[:-3] for k in z.files if k.endswith("_in")]
    assert len(names) >= 5
    for n in names:
        want = z[n + "_out"]
        got = resize_ref.resize_bicubic_u8(z[n + "_in"], want.shape[0], want.shape[1])
        assert np.array_equal(got, want), n


def test_oracle_equals_pillow_live():
    Image = pytest.importorskip("PIL.Image")
    rs = np.random.RandomState(3)
    for (h, w, oh, ow) in [(360, 640, 224, 224), (240, 320, 224, 224), (224, 398, 224, 224), (100, 224, 224, 224),
                           (33, 47, 64, 16)]:
        x = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
        want = np.array(Image.fromarray(x).resize((ow, oh)))
        assert np.array_equal(resize_ref.resize_bicubic_u8(x, oh, ow), want), (h, w, oh, ow)


def test_host_coefficient_tables_equal_the_oracle():
    from vidsitu_amd import _lib

    lib = _lib.load()
    for (i, o) in [(640, 224), (360, 224), (398, 224), (12, 24), (224, 224), (1920, 224), (37, 20), (225, 224)]:
        ks = lib.vs_resize_ksize(i, o)
        b = torch.zeros((o, 2), dtype=torch.int32)
        k = torch.zeros((o, ks), dtype=torch.int32)
        assert lib.vs_resize_coeffs(i, o, C.c_void_p(b.data_ptr()), C.c_void_p(k.data_ptr())) == 0
        ks2, b2, k2 = resize_ref.precompute_coeffs(i, o)
        assert ks == ks2 and np.array_equal(b.numpy(), b2) and np.array_equal(k.numpy(), k2), (i, o)
