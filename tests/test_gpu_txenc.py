"""The HIP TxEncoder (vidsitu_amd/transformer_code.py through the C-ABI) against the golden
vectors produced by the REFERENCE's own module, forward and backward, plus the small fp32
ops it is built from.  fp32 kernels, different summation order: 2e-4 of max magnitude."""
import glob
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gpu_utils import assert_close

pytestmark = pytest.mark.gpu

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "txenc_*.npz")))
TOL = 2e-4


def _load(path, dev):
    from oracle.txenc_ref import make_weights
    from vidsitu_amd.transformer_code import Transformer

    g = np.load(path)
    d, dh, nl, nh, B, L, seed = [int(v) for v in g["cfg"]]
    mdl = Transformer(d_model=d, n_vocab_src=0, vocab_trg=0, d_hidden=dh, n_layers=nl, n_heads=nh,
                      drop_ratio=0.1, pe=False)
    sd = {"encoder." + k: torch.from_numpy(v) for k, v in make_weights(d, dh, nl, seed).items()}
    mdl.load_state_dict(sd, strict=True)  # reference key names load unchanged
    return g, mdl.to(dev).eval()


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_txenc_forward_matches_reference_golden(path, dev):
    g, mdl = _load(path, dev)
    with torch.no_grad():
        y = mdl.encoder(torch.from_numpy(g["x"]).to(dev))[-1]
    assert_close(y, torch.from_numpy(g["y"]), TOL, "encoder output")


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_txenc_backward_matches_reference_golden(path, dev):
    g, mdl = _load(path, dev)
    x = torch.from_numpy(g["x"]).to(dev).requires_grad_()
    y = mdl.encoder(x)[-1]
    y.backward(torch.from_numpy(g["dy"]).to(dev))
    assert_close(x.grad, torch.from_numpy(g["dx"]), 5e-4, "dx")
    for k, p in mdl.encoder.named_parameters():
        if p.grad.ndim == 1:
            assert_close(p.grad, torch.from_numpy(g["g." + k]), 5e-4, k)
        else:
            assert_close(p.grad[:32, :32], torch.from_numpy(g["gc." + k]), 5e-4, k)
            s = g["gs." + k]
            assert abs(float(p.grad.double().abs().sum()) - s[1]) <= 1e-3 * s[1], k


@pytest.mark.parametrize("m,n,k,relu", [(1, 7, 5, False), (8, 1564, 1152, False), (40, 1024, 1024, True),
                                        (40, 1024, 2304, True), (50, 33, 130, False), (130, 64, 96, True),
                                        # whole-x-in-LDS kernel (M <= 16, K <= 2048, K % 4 == 0)
                                        (8, 1024, 1024, True), (16, 515, 2048, False), (3, 40, 12, True),
                                        (10, 1024, 2304, False), (8, 1024, 2304, True), (5, 77, 4096, False)])
def test_linear_fwd_bwd(m, n, k, relu, dev):
    from vidsitu_amd import ops

    g = torch.Generator().manual_seed(m * 7 + n)
    x = torch.randn(m, k, generator=g).requires_grad_()
    w = (torch.randn(n, k, generator=g) / k ** 0.5).requires_grad_()
    b = torch.randn(n, generator=g).requires_grad_()
    yr = F.linear(x, w, b)
    yr = F.relu(yr) if relu else yr
    y = ops.linear_fwd(x.detach().to(dev), w.detach().to(dev), b.detach().to(dev), relu)
    assert_close(y, yr, TOL, "linear fwd")
    dy = torch.randn(m, n, generator=g)
    dy_eff = dy * (yr > 0) if relu else dy
    gx, gw, gb = torch.autograd.grad(yr, [x, w, b], dy)
    dx, dw, db = ops.linear_bwd(dy_eff.to(dev), x.detach().to(dev), w.detach().to(dev))
    assert_close(dx, gx, TOL, "dx")
    assert_close(dw, gw, TOL, "dw")
    assert_close(db, gb, TOL, "db")


@pytest.mark.parametrize("B,L,H,dh", [(2, 5, 8, 128), (3, 5, 8, 8), (1, 16, 2, 40), (2, 1, 4, 16)])
def test_attention_small(B, L, H, dh, dev):
    from vidsitu_amd import ops

    g = torch.Generator().manual_seed(B + L)
    q, k, v = [torch.randn(B, L, H * dh, generator=g).requires_grad_() for _ in range(3)]
    scale = (H * dh) ** 0.5
    outs = []
    for h in range(H):
        sl = slice(h * dh, (h + 1) * dh)
        p = F.softmax(q[..., sl] @ k[..., sl].transpose(1, 2) / scale, -1)
        outs.append(p @ v[..., sl])
    ref = torch.cat(outs, -1)
    o, probs = ops.attn_small_fwd(q.detach().to(dev), k.detach().to(dev), v.detach().to(dev), H, scale)
    assert_close(o, ref, TOL, "attention out")
    do = torch.randn(ref.shape, generator=g)
    gq, gk, gv = torch.autograd.grad(ref, [q, k, v], do)
    dq, dk, dv = ops.attn_small_bwd(q.detach().to(dev), k.detach().to(dev), v.detach().to(dev), probs,
                                    do.to(dev), H, scale)
    assert_close(dq, gq, TOL, "dq")
    assert_close(dk, gk, TOL, "dk")
    assert_close(dv, gv, TOL, "dv")


@pytest.mark.parametrize("rows,D", [(40, 1024), (3, 64), (5, 2048), (7, 100)])
def test_add_layernorm(rows, D, dev):
    from vidsitu_amd import ops

    g = torch.Generator().manual_seed(rows)
    x, r = torch.randn(rows, D, generator=g).requires_grad_(), torch.randn(rows, D, generator=g).requires_grad_()
    gamma, beta = (torch.rand(D, generator=g) + 0.5).requires_grad_(), torch.randn(D, generator=g).requires_grad_()
    ref = F.layer_norm(x + r, (D,), gamma, beta, 1e-5)
    y, mean, rstd = ops.add_layernorm_fwd(x.detach().to(dev), r.detach().to(dev), gamma.detach().to(dev),
                                          beta.detach().to(dev))
    assert_close(y, ref, TOL, "layernorm")
    dy = torch.randn(rows, D, generator=g)
    gx, gr, gg, gb = torch.autograd.grad(ref, [x, r, gamma, beta], dy)
    dx, dr, dg, db = ops.add_layernorm_bwd(dy.to(dev), x.detach().to(dev), r.detach().to(dev),
                                           gamma.detach().to(dev), mean, rstd)
    assert_close(dx, gx, TOL, "dx")
    assert_close(dr, gr, TOL, "dr")
    assert_close(dg, gg, TOL, "dgamma")
    assert_close(db, gb, TOL, "dbeta")


def test_dropout_masks_in_attention_and_residual(dev):
    """Train-mode dropout sites of transformer_code.py (:48 on the attention probabilities,
    :30 on the residual branch) with explicit masks vs the same masks applied in torch."""
    from vidsitu_amd import ops

    g = torch.Generator().manual_seed(3)
    B, L, H, dh, p = 2, 5, 4, 16, 0.25
    q, k, v = [torch.randn(B, L, H * dh, generator=g).requires_grad_() for _ in range(3)]
    mask = (torch.rand(B, H, L, L, generator=g) >= p).float() / (1 - p)
    scale = (H * dh) ** 0.5
    outs = []
    for h in range(H):
        sl = slice(h * dh, (h + 1) * dh)
        pr = F.softmax(q[..., sl] @ k[..., sl].transpose(1, 2) / scale, -1) * mask[:, h]
        outs.append(pr @ v[..., sl])
    ref = torch.cat(outs, -1)
    o, probs = ops.attn_small_fwd(q.detach().to(dev), k.detach().to(dev), v.detach().to(dev), H, scale,
                                  mask.to(dev))
    assert_close(o, ref, TOL, "attention with dropout")
    do = torch.randn(ref.shape, generator=g)
    gq, gk, gv = torch.autograd.grad(ref, [q, k, v], do)
    dq, dk, dv = ops.attn_small_bwd(q.detach().to(dev), k.detach().to(dev), v.detach().to(dev), probs,
                                    do.to(dev), H, scale, mask.to(dev))
    assert_close(dq, gq, TOL, "dq")
    assert_close(dk, gk, TOL, "dk")
    assert_close(dv, gv, TOL, "dv")
    rows, D = 10, 64
    x, r = torch.randn(rows, D, generator=g).requires_grad_(), torch.randn(rows, D, generator=g).requires_grad_()
    gamma, beta = (torch.rand(D, generator=g) + 0.5), torch.randn(D, generator=g)
    rmask = (torch.rand(rows, D, generator=g) >= p).float() / (1 - p)
    ref = F.layer_norm(x + r * rmask, (D,), gamma, beta, 1e-5)
    y, mean, rstd = ops.add_layernorm_fwd(x.detach().to(dev), r.detach().to(dev), gamma.to(dev),
                                          beta.to(dev), 1e-5, rmask.to(dev))
    assert_close(y, ref, TOL, "LN(x + dropout(r))")
    dy = torch.randn(rows, D, generator=g)
    gx, gr = torch.autograd.grad(ref, [x, r], dy)
    dx, dr, _, _ = ops.add_layernorm_bwd(dy.to(dev), x.detach().to(dev), r.detach().to(dev), gamma.to(dev),
                                         mean, rstd, rmask.to(dev))
    assert_close(dx, gx, TOL, "dx")
    assert_close(dr, gr, TOL, "dr (masked)")


def test_txenc_train_mode_dropout_statistics(dev):
    """Encoder in train mode: runs end to end with dropout 0.1, is stochastic, differentiable,
    and equals the eval output when the masks are forced to one."""
    from vidsitu_amd import ops
    from vidsitu_amd.transformer_code import Transformer

    torch.manual_seed(0)
    mdl = Transformer(d_model=64, n_vocab_src=0, vocab_trg=0, d_hidden=64, n_layers=2, n_heads=8,
                      drop_ratio=0.1).to(dev)
    x = torch.randn(3, 5, 64, device=dev, requires_grad=True)
    mdl.train()
    a, b = mdl(x), mdl(x)
    assert float((a - b).abs().max()) > 1e-3  # different masks
    a.sum().backward()
    assert x.grad is not None and torch.isfinite(x.grad).all()
    keep = ops.dropout_mask
    try:
        ops.dropout_mask = lambda shape, p, device: torch.ones(shape, device=device)
        c = mdl(x)
    finally:
        ops.dropout_mask = keep
    mdl.eval()
    assert_close(c, mdl(x), 1e-6, "train mode with unit masks == eval")


def test_softmax_xent_and_topk(dev):
    from vidsitu_amd import ops

    g = torch.Generator().manual_seed(0)
    logits = (torch.randn(40, 1564, generator=g) * 2).requires_grad_()
    labels = torch.randint(0, 1564, (40,), generator=g)
    ref = F.cross_entropy(logits, labels)
    (gref,) = torch.autograd.grad(ref, logits)
    loss, dl = ops.softmax_xent(logits.detach().to(dev), labels.to(dev))
    assert abs(float(loss) - float(ref)) < 1e-5 * abs(float(ref))
    assert_close(dl, gref, 1e-4, "dlogits")
    # EvalB: softmax -> sort(descending) -> first 5 (evl_vsitu.py:39-42): indices bit-exact
    probs, idx = ops.softmax_topk(logits.detach().to(dev), 5)
    ps, ix = F.softmax(logits.detach(), -1).sort(dim=-1, descending=True)
    assert torch.equal(idx.cpu(), ix[:, :5])
    assert_close(probs, ps[:, :5], 1e-5, "top-5 probabilities")
    # ties resolve to the lowest index, deterministically
    t = torch.zeros(2, 10)
    t[0, [3, 7]] = 1.0
    _, ti = ops.softmax_topk(t.to(dev), 3)
    assert ti.cpu().tolist() == [[3, 7, 0], [0, 1, 2]]


def test_adam_matches_torch(dev):
    from vidsitu_amd import ops

    g = torch.Generator().manual_seed(1)
    p0 = torch.randn(10007, generator=g)
    pr = p0.clone().requires_grad_()
    opt = torch.optim.Adam([pr], lr=1e-3, betas=(0.9, 0.99))
    p = p0.to(dev)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for step in range(1, 4):
        gr = torch.randn(10007, generator=g)
        pr.grad = gr.clone()
        opt.step()
        ops.adam_step(p, gr.to(dev), m, v, 1e-3, 0.9, 0.99, 1e-8, step)
    assert_close(p, pr.detach(), 1e-6, "adam params after 3 steps")
    # device-side step counter variant (what a captured hipGraph replays)
    p2 = p0.to(dev)
    m2, v2, cnt = torch.zeros_like(p2), torch.zeros_like(p2), torch.zeros(1, dtype=torch.int32, device=dev)
    g2 = torch.Generator().manual_seed(1)
    torch.randn(10007, generator=g2)
    for _ in range(3):
        ops.adam_step_dev(p2, torch.randn(10007, generator=g2).to(dev), m2, v2, 1e-3, 0.9, 0.99, 1e-8, cnt)
    assert int(cnt) == 3
    assert_close(p2, pr.detach(), 1e-6, "adam (device step counter)")


def test_adam_with_fused_bf16_cast_and_arena_refresh(dev):
    """vs_adam_step_dev_cast == vs_adam_step_dev + vs_cast_f32_to_bf16 (bitwise), device step
    counter shared."""
    from vidsitu_amd import ops

    g = torch.Generator().manual_seed(3)
    n = 4096 + 8
    p0 = torch.randn(n, generator=g).to(dev)
    gr = torch.randn(n, generator=g).to(dev)
    pa, pb = p0.clone(), p0.clone()
    ma, va, mb, vb = (torch.zeros(n, device=dev) for _ in range(4))
    ta = torch.zeros(1, dtype=torch.int32, device=dev)
    tb = torch.zeros(1, dtype=torch.int32, device=dev)
    bf = torch.zeros(n, dtype=torch.bfloat16, device=dev)
    ref_bf = torch.zeros(n, dtype=torch.bfloat16, device=dev)
    for _ in range(3):
        ops.adam_step_dev(pa, gr, ma, va, 1e-2, 0.9, 0.99, 1e-8, ta, grad_scale=0.5)
        ops.adam_step_dev_cast(pb, gr, mb, vb, bf, 1e-2, 0.9, 0.99, 1e-8, tb, grad_scale=0.5)
    ops.cast_bf16(pa, ref_bf)
    assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)
    assert torch.equal(bf.view(torch.int16), ref_bf.view(torch.int16)) and int(tb) == 3


@pytest.mark.parametrize("arena_on", [True, False], ids=["arena", "loose"])
def test_residual_gradient_joins_in_the_linear_epilogue_bitwise(arena_on, dev, monkeypatch):
    """`ResidualBlock.route_grads`: the LayerNorm's dx reaches the wrapped layer's first op through a side channel and
    is added in that op's data-gradient epilogue (`vs_linear_bwd_fused_res`) instead of by an add launch of
    autograd's: input gradient and parameter gradients bit for bit, with and without a parameter arena (without one
    the q / k / v projections are three ops and the attention block keeps autograd's add)."""
    from vidsitu_amd import transformer_code as T
    from vidsitu_amd.optim import ParamArena

    d, layers, b, l = 1024, 3, 2, 4
    torch.manual_seed(5)
    mdl = T.Transformer(d_model=d, n_vocab_src=0, vocab_trg=0, d_hidden=d, n_layers=layers, n_heads=8,
                        drop_ratio=0.1, pe=False).to(dev).train()
    if arena_on:
        ParamArena(mdl)
    x0, dy = torch.randn(b, l, d, device=dev), torch.randn(b, l, d, device=dev)

    def run(route):
        monkeypatch.setattr(T.ResidualBlock, "route_grads", route)
        n_adds = []
        orig = T.ops.linear_bwd
        monkeypatch.setattr(T.ops, "linear_bwd", lambda *a, **k: (n_adds.append(k.get("dx_res") is not None), orig(*a, **k))[1])
        for p in mdl.parameters():
            if p.grad is not None:
                p.grad.fill_(float("nan")) if arena_on else None
            if not arena_on:
                p.grad = None
        torch.manual_seed(11)
        T._masks.__init__()
        x = x0.clone().requires_grad_()
        out = mdl.encoder(x)[-1]
        out.backward(dy)
        torch.cuda.synchronize()
        monkeypatch.setattr(T.ops, "linear_bwd", orig)
        return out.detach().clone(), x.grad.clone(), [p.grad.clone() for p in mdl.parameters()], sum(n_adds)

    o0, dx0, g0, n0 = run(False)
    o1, dx1, g1, n1 = run(True)
    assert n0 == 0 and n1 == (2 if arena_on else 1) * layers
    assert torch.equal(o0, o1) and torch.equal(dx0, dx1)
    bad = [k for (k, _), a, c in zip(mdl.named_parameters(), g0, g1) if not torch.equal(a, c)]
    assert not bad, bad


