"""The TxEncoder oracle restatement vs golden vectors produced by the REFERENCE's own
module (tests/golden/gen_txenc_golden.py).  CPU only."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle.txenc_ref import encoder_forward, make_weights

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "txenc_*.npz")))


def test_fixtures_present():
    assert len(GOLD) >= 5


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_oracle_matches_reference_forward(path):
    g = np.load(path)
    d, dh, nl, nh, B, L, seed = [int(v) for v in g["cfg"]]
    w = make_weights(d, dh, nl, seed)
    y = encoder_forward(g["x"], w, nl, nh).numpy()
    # same fp32 arithmetic in the same order: bit-exact on this torch build
    np.testing.assert_allclose(y, g["y"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("path", GOLD[:3], ids=[os.path.basename(p) for p in GOLD[:3]])
def test_oracle_matches_reference_backward(path):
    g = np.load(path)
    d, dh, nl, nh, B, L, seed = [int(v) for v in g["cfg"]]
    w = {k: torch.from_numpy(v).requires_grad_() for k, v in make_weights(d, dh, nl, seed).items()}
    x = torch.from_numpy(g["x"]).requires_grad_()
    from oracle.txenc_ref import encoder_layer

    h = x
    for i in range(nl):
        h = encoder_layer(h, w, f"layers.{i}.", nh)
    h.backward(torch.from_numpy(g["dy"]))
    np.testing.assert_allclose(x.grad.numpy(), g["dx"], rtol=1e-5, atol=1e-6)
    for k, p in w.items():
        if p.grad.ndim == 1:
            np.testing.assert_allclose(p.grad.numpy(), g["g." + k], rtol=1e-4, atol=1e-5)
        else:
            np.testing.assert_allclose(p.grad.numpy()[:32, :32], g["gc." + k], rtol=1e-4, atol=1e-5)


def test_scale_is_sqrt_d_model_not_head_dim():
    """Parity trap (transformer_code.py:36,54): with the head-dim scale the golden fails."""
    g = np.load(GOLD[0])
    d, dh, nl, nh, B, L, seed = [int(v) for v in g["cfg"]]
    import math
    import oracle.txenc_ref as ref

    w = make_weights(d, dh, nl, seed)
    y = encoder_forward(g["x"], w, nl, nh).numpy()
    assert np.abs(y - g["y"]).max() < 1e-6
    orig = math.sqrt
    try:
        ref.math = type("M", (), {"sqrt": staticmethod(lambda v: orig(v / nh))})
        y_bad = encoder_forward(g["x"], w, nl, nh).numpy()
    finally:
        ref.math = math
    assert np.abs(y_bad - g["y"]).max() > 1e-3
