"""Non-local block (slowfast `Nonlocal`, i3d_r50_nl_8x8) on the HIP kernels against
the fp32 torch oracle (`oracle.slowfast_ref.Nonlocal`, parity unpinned like the rest of the SlowFast
oracle): forward in train and eval mode, the gradient of the input and of every parameter; the helper
kernels one by one; the whole I3D-NL trunk.  Activations (scores and probabilities included) are bf16:
tolerances are relative L2 errors of whole tensors -- 2e-2 forward, 5e-2 gradients."""
import pytest
import torch

from gpu_utils import rel_l2

pytestmark = pytest.mark.gpu


def _rb(t):
    return t.to(torch.bfloat16).float()


def test_maxpool_hw2_softmax_rows_colsum(dev):
    from vidsitu_amd import ops

    g = torch.Generator().manual_seed(0)
    x = _rb(torch.randn(2, 16, 3, 6, 10, generator=g))
    x[0, :, 0, 0, 0] = x[0, :, 0, 0, 1]  # a tie: the first maximum wins
    xd = x.to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last_3d)
    y, idx = ops.maxpool_hw2(xd)
    ref, ridx = torch.nn.functional.max_pool3d(x, (1, 2, 2), (1, 2, 2), return_indices=True)
    assert torch.equal(y.float().cpu(), ref)
    dy = _rb(torch.randn(ref.shape, generator=g))
    dyd = dy.to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last_3d)
    dx = ops.maxpool_hw2_bwd(dyd, idx, tuple(x.shape))
    xr = x.clone().requires_grad_(True)
    torch.nn.functional.max_pool3d(xr, (1, 2, 2), (1, 2, 2)).backward(dy)
    assert torch.equal(dx.float().cpu(), xr.grad)
    # row softmax and its backward (bf16 storage, fp32 math)
    rows, p = 37, 1568
    s = _rb(torch.randn(rows, p, generator=g) * 3)
    sd = s.to(dev).to(torch.bfloat16)
    prob = ops.softmax_rows_bf16(sd.clone(), rows, p).float().cpu()
    want = torch.softmax(s, dim=1)
    assert float((prob - want).abs().max()) < 2 ** -8 * float(want.max())
    dp = _rb(torch.randn(rows, p, generator=g))
    pb = _rb(want)
    ds = ops.softmax_rows_bwd_bf16(pb.to(dev).to(torch.bfloat16), dp.to(dev).to(torch.bfloat16).clone(), rows, p,
                                   0.25).float().cpu()
    want_ds = 0.25 * pb * (dp - (dp * pb).sum(1, keepdim=True))
    assert rel_l2(ds, want_ds) < 1e-2
    cs = ops.colsum_bf16(xd).cpu()
    assert torch.allclose(cs, x.sum(dim=(0, 2, 3, 4)), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("inst,pool,thw", [("softmax", [1, 2, 2], (4, 8, 8)), ("dot_product", [1, 2, 2], (4, 8, 8)),
                                           ("softmax", None, (4, 8, 8)),
                                           # 4 x 7 x 7 = 196 key positions (res4 of the real model): padded to 200
                                           ("softmax", [1, 2, 2], (4, 14, 14)), ("dot_product", [1, 2, 2], (4, 14, 14))])
def test_nonlocal_block_matches_oracle(inst, pool, thw, dev):
    from types import SimpleNamespace

    from oracle import slowfast_ref as R
    from vidsitu_amd import ops
    from vidsitu_amd.trunk import Conv3dP, Nonlocal

    torch.manual_seed(1)
    dim, inner, n = 64, 32, 2
    t, h, w = thw
    cfg = SimpleNamespace(BN=SimpleNamespace(EPSILON=1e-5, MOMENTUM=0.1))
    ref = R.Nonlocal(dim, inner, pool, inst, cfg)
    with torch.no_grad():
        for m in (ref.conv_theta, ref.conv_phi, ref.conv_g, ref.conv_out):
            m.weight.copy_(_rb(torch.randn_like(m.weight) * (2.0 / m.out_channels) ** 0.5))
            m.bias.copy_(torch.randn_like(m.bias) * 0.2)
        ref.bn.weight.copy_(0.5 + torch.rand(dim))
        ref.bn.bias.copy_(torch.randn(dim) * 0.1)
        ref.bn.running_mean.copy_(torch.randn(dim) * 0.1)
        ref.bn.running_var.copy_(0.5 + torch.rand(dim))
    ours = Nonlocal(dim, inner, pool, inst, 1e-5, 0.1)
    ours.load_state_dict(ref.state_dict(), strict=True)
    ours = ours.to(dev)
    for m in ours.modules():
        if isinstance(m, Conv3dP):
            m.refresh()
    x = _rb(torch.randn(n, dim, t, h, w))
    xd = x.to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last_3d)
    # ---- eval: running statistics folded, conv_out's bias in the shift
    ref.eval()
    with torch.no_grad():
        want = ref(x)
        sc, sh, _, _ = ops.bn_finalize(None, 0, ours.bn.weight, ours.bn.bias, ours.bn.running_mean,
                                       ours.bn.running_var, ours.bn.momentum, ours.bn.eps, train=False)
        ours.bn.fold = (sc, sh)
        got = ours.fwd(xd, None, False, None)
    assert rel_l2(got.float().cpu() - x, want - x) < 2e-2  # the branch itself, not branch + identity
    # ---- train: batch statistics, backward
    ref.train()
    xr = x.clone().requires_grad_(True)
    out_r = ref(xr)
    gout = _rb(torch.randn(out_r.shape) / out_r.numel() ** 0.5)
    (out_r * gout).sum().backward()
    saved = []
    out_o = ours.fwd(xd, None, True, saved)
    assert rel_l2(out_o.float().cpu() - x, out_r.detach() - x) < 2e-2
    assert rel_l2(ours.bn.running_mean.cpu(), ref.bn.running_mean) < 1e-2
    gd = gout.to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last_3d)
    dx = ours.bwd(saved, gd)
    assert not saved
    torch.cuda.synchronize()
    assert rel_l2(dx.float().cpu() - gout, xr.grad - gout) < 5e-2  # branch gradient
    pr, po = dict(ref.named_parameters()), dict(ours.named_parameters())
    for k, v in pr.items():
        g = po[k].grad.float().cpu()
        if k == "conv_out.bias":  # removed by the batch norm behind it: exactly zero here, noise in autograd
            assert float(g.abs().max()) == 0.0 and float(v.grad.abs().max()) < 1e-5
            continue
        # conv_phi.bias has an exactly-zero true gradient under the softmax (a constant added to every key
        # shifts all scores of a query equally): both sides hold rounding noise there -- bf16 noise on ours
        # -- so a bias is measured against at least 5 % of its conv's weight-gradient norm
        floor = 0.05 * float(pr[k.replace(".bias", ".weight")].grad.norm()) if k.endswith(".bias") else 0.0
        err = float((g - v.grad).norm()) / max(float(v.grad.norm()), floor, 1e-20)
        assert err < 5e-2, (k, err)


def test_i3d_nl_trunk_eval_matches_oracle_and_trains(dev):
    from oracle.slowfast_ref import VideoTrunk as RefTrunk, default_sf_cfg, randomize_bn
    from vidsitu_amd.trunk import VideoTrunk

    torch.manual_seed(0)
    cfg = default_sf_cfg("i3d", "mini", 8, 8)
    cfg.NONLOCAL.LOCATION = [[[]], [[0]], [[0]], [[]]]
    cfg.NONLOCAL.INSTANTIATION = "softmax"
    ref = RefTrunk(cfg)
    randomize_bn(ref, 0)
    with torch.no_grad():
        for name, m in ref.named_modules():  # non-trivial non-local branches (gamma is 0 at init)
            if name.endswith("_nonlocal0"):
                m.bn.weight.copy_(0.3 + 0.4 * torch.rand(m.bn.num_features))
                for c in (m.conv_theta, m.conv_phi, m.conv_g, m.conv_out):
                    c.bias.copy_(torch.randn_like(c.bias) * 0.1)
    ours = VideoTrunk(cfg)
    ours.load_state_dict(ref.state_dict(), strict=True)
    ours = ours.to(dev)
    x = torch.randn(2, 3, 8, 64, 64)
    ref.eval(), ours.eval()
    with torch.no_grad():
        fr = ref.forward_features([x])[0]
        fo = ours.forward_features([x.to(dev)])[0]
    assert rel_l2(fo.float().cpu(), fr) < 3e-2
    # a train step through the manual backward: every parameter receives a finite gradient
    ours.train()
    fo = ours.forward_features([x.to(dev)])[0]
    fo.float().square().mean().backward()
    for k, p in ours.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
    assert any("nonlocal" in k and float(p.grad.abs().max()) > 0 for k, p in ours.named_parameters())
