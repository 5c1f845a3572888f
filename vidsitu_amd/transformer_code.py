"""Host-side mirror of the reference's `utils/transformer_code.py` (post-LN
transformer encoder, lines 21-124 and 261-284) on the HIP kernels.

Same class names, constructor arguments, attribute names and state_dict keys
(`encoder.layers.{i}.selfattn.layer.wq.weight`, ...), so a reference checkpoint
loads unchanged; `forward` never calls torch.nn.functional -- every matmul,
softmax and LayerNorm is a kernel of libvidsitu_hip.so (fp32, exact-order
reductions).  Parity trap kept: the softmax scale is sqrt(d_model), not
sqrt(head_dim) (`utils/transformer_code.py:36,54`).
"""
import math
import os

import torch
from torch import nn

from . import ops


class _Lazy:
    """LayerNorm launches parked for the linear that consumes them (`Encoder.forward` and its backward only).

    On the encoder's 8 token rows every launch is ~11 us of the step's critical path and a LayerNorm is a microsecond of
    work: `AddLayerNormFn.forward` allocates its outputs and parks the launch; the next `LinearFn` / `FusedQKVAttnFn`
    whose input IS that output runs LayerNorm + linear as one kernel (`ops.ln_linear_fwd`: every block recomputes the
    8 rows in its prologue, block 0 stores them).  Backward: `AddLayerNormFn.backward` parks, the backward of the linear
    that receives its `dr` runs both (`ops.ln_bwd_linear_bwd`).  Anything else that could read a parked output flushes
    it first (the stand-alone launch into the same tensors): same bits either way.
    OFF by default (VS_LN_LINEAR_FUSE=1 switches it on): measured SLOWER -- train step 12.47-12.63 vs 12.11-12.17 ms;
    per launch (tools/ln_linear_time.py, dependent chains in a hipGraph) LN ; linear = 9.3 us against 10.9 us fused
    (forward) and 9.5 against 16.3 us (backward of the 2048-wide feed-forward): inside one kernel the LayerNorm's
    dependent loads run IN FRONT of the weight stream instead of beside the previous kernel's tail, and the weight-
    gradient blocks read the LayerNorm's result through flat loads from LDS.  profiles/r03_encoder_section.txt."""

    enabled = os.environ.get("VS_LN_LINEAR_FUSE", "0") == "1"
    active = False
    fwd = None
    bwd = None
    fused = [0, 0]  # forward / backward launches that took a parked LayerNorm (tests)

    @staticmethod
    def flush_fwd():
        p, _Lazy.fwd = _Lazy.fwd, None
        if p is not None:
            rows, d = p["x"].shape
            ops._lib.call("vs_add_layernorm_fwd", ops._ptr(p["x"]), ops._ptr(p["r"]), ops._ptr(p["rmask"]),
                          ops._ptr(p["gamma"]), ops._ptr(p["beta"]), ops._ptr(p["y"]), ops._ptr(p["mean"]),
                          ops._ptr(p["rstd"]), rows, d, float(p["eps"]), ops._stream())

    @staticmethod
    def take_fwd(x2, w, b):
        """The parked LayerNorm whose output x2 is, if the fused kernel takes this linear; else flush and None."""
        p = _Lazy.fwd
        if p is None:
            return None
        if (p["y"].data_ptr() == x2.data_ptr() and x2.is_contiguous() and w.dtype == torch.float32
                and w.is_contiguous() and w.data_ptr() % 16 == 0 and (b is None or b.dtype == torch.float32)):
            _Lazy.fwd = None
            return p
        _Lazy.flush_fwd()
        return None

    @staticmethod
    def flush_bwd():
        p, _Lazy.bwd = _Lazy.bwd, None
        if p is not None:
            rows, d = p["x"].shape
            ops._lib.call("vs_add_layernorm_bwd", ops._ptr(p["dy"]), ops._ptr(p["x"]), ops._ptr(p["r"]),
                          ops._ptr(p["rmask"]), ops._ptr(p["gamma"]), ops._ptr(p["mean"]), ops._ptr(p["rstd"]),
                          ops._ptr(p["dx"]), ops._ptr(p["dr"]) if p["rmask"] is not None else None, ops._ptr(p["dg"]),
                          ops._ptr(p["db"]), rows, d, ops._stream())


class LinearFn(torch.autograd.Function):
    """y = act(x @ W^T + b) on vs_linear_*  (x: [..., K] fp32)."""

    @staticmethod
    def forward(ctx, x, w, b, relu, route):
        ctx.route = route  # see ResidualBlock: the residual path's gradient joins this layer's dx in the kernel
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        pend = _Lazy.take_fwd(x2, w, b)
        if pend is not None:  # LayerNorm + this linear as one launch
            y = ops.ln_linear_fwd(pend["x"], pend["r"], pend["gamma"], pend["beta"], pend["eps"], pend["rmask"],
                                  pend["y"], pend["mean"], pend["rstd"], w, b, relu)
            _Lazy.fused[0] += 1
        else:
            y = ops.linear_fwd(x2, w, b, relu)
        ctx.save_for_backward(x2, w, y if relu else None)
        ctx.has_bias, ctx.relu, ctx.shp = b is not None, relu, shp
        # parameters that live in a ParamArena take their gradient in place (overwrite
        # semantics, like the conv / BN parameters) instead of through autograd's accumulate
        ctx.w_param = w if getattr(w, "_vs_direct_grad", False) else None
        # the arena's [K][N] image of w, unless w was modified behind the arena's back since
        ctx.w_wt = w._vs_wt if getattr(w, "_vs_wt_version", None) == w._version else None
        ctx.b_param = b if (b is not None and getattr(b, "_vs_direct_grad", False)) else None
        return y.reshape(*shp[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w, y = ctx.saved_tensors
        dy2 = dy.reshape(-1, w.shape[0])
        direct = ctx.w_param is not None and ctx.w_param.grad is not None and \
            (not ctx.has_bias or (ctx.b_param is not None and ctx.b_param.grad is not None))
        wt = ctx.w_wt  # the arena's transposed image (refreshed once per step)
        if wt is not None:
            from .trunk import Conv3dP

            if Conv3dP._pending_arenas:  # an asynchronous refresh may still be in flight
                Conv3dP.join_pending_refresh()
        pend = _Lazy.bwd
        if pend is not None:
            if (pend["dr"].data_ptr() == dy2.data_ptr() and direct and wt is not None and not ctx.relu
                    and ctx.needs_input_grad[0] and wt.data_ptr() % 16 == 0 and x2.data_ptr() % 16 == 0
                    and x2.shape[1] % 4 == 0 and (ctx.route is None or "dx" not in ctx.route)):
                _Lazy.bwd = None  # the LayerNorm's backward + both gradients of this linear as one launch
                dx = ops.ln_bwd_linear_bwd(pend["dy"], pend["x"], pend["r"], pend["gamma"], pend["mean"], pend["rstd"],
                                           pend["rmask"], pend["dx"], pend["dg"], pend["db"], x2, wt,
                                           ctx.w_param.grad, ctx.b_param.grad if ctx.has_bias else None)
                _Lazy.fused[1] += 1
                return dx.reshape(ctx.shp), None, None, None, None
            _Lazy.flush_bwd()
        dx, dw, db = ops.linear_bwd(
            dy2.contiguous(), x2, w, need_dx=ctx.needs_input_grad[0], has_bias=ctx.has_bias,
            dw_out=ctx.w_param.grad if direct else None,
            db_out=ctx.b_param.grad if (direct and ctx.has_bias) else None, wt=wt,
            relu_y=y if ctx.relu else None,  # the ReLU mask is applied inside the gradient kernels
            dx_res=ctx.route.pop("dx", None) if ctx.route is not None else None)
        dx = dx.reshape(ctx.shp) if dx is not None else None
        if direct:
            return dx, None, None, None, None
        return dx, dw, db, None, None


class AttnSmallFn(torch.autograd.Function):
    """concat_h softmax(Q_h K_h^T / scale) V_h for L <= 16 (vs_attn_small_*)."""

    @staticmethod
    def forward(ctx, q, k, v, n_heads, scale, drop_mask):
        _Lazy.flush_fwd()
        o, probs = ops.attn_small_fwd(q, k, v, n_heads, scale, drop_mask)
        ctx.save_for_backward(q, k, v, probs, drop_mask)
        ctx.n_heads, ctx.scale = n_heads, scale
        return o

    @staticmethod
    def backward(ctx, do):
        _Lazy.flush_bwd()
        q, k, v, probs, drop_mask = ctx.saved_tensors
        dq, dk, dv = ops.attn_small_bwd(q.contiguous(), k.contiguous(), v.contiguous(), probs, do,
                                        ctx.n_heads, ctx.scale, drop_mask)
        return dq, dk, dv, None, None, None


class FusedQKVAttnFn(torch.autograd.Function):
    """Self-attention of one `MultiHead` as ONE autograd node: the q / k / v projections run as one GEMM on
    the [3d, d] block the three weights form in the parameter arena (`ParamArena._adopt_linear_weights`
    sets `MultiHead._qkv`), the attention kernels read / write the fused [rows, 3d] buffers in place.
    Per layer this replaces 3 forward launches, 3 + 3 backward launches and two gradient adds by 1 + 1 + 1
    -- the encoder runs on 8 tokens, every launch is pure latency on the critical path of the step."""

    @staticmethod
    def forward(ctx, x, fused, n_heads, scale, drop_mask, route):
        ctx.route = route
        b, l, d = x.shape
        x2 = x.reshape(b * l, d)
        pend = _Lazy.take_fwd(x2, fused["w"], None)
        if pend is not None:  # the previous block's LayerNorm + the q | k | v projection as one launch
            qkv = ops.ln_linear_fwd(pend["x"], pend["r"], pend["gamma"], pend["beta"], pend["eps"], pend["rmask"],
                                    pend["y"], pend["mean"], pend["rstd"], fused["w"], None, False)
            _Lazy.fused[0] += 1
        else:
            qkv = ops.linear_fwd(x2, fused["w"], None, False)
        o, probs = ops.attn_small_fwd_fused(qkv, b, l, n_heads, scale, drop_mask)
        ctx.save_for_backward(x2, qkv, probs, drop_mask)
        ctx.fused, ctx.n_heads, ctx.scale, ctx.bl = fused, n_heads, scale, (b, l)
        return o

    @staticmethod
    def backward(ctx, do):
        _Lazy.flush_bwd()
        x2, qkv, probs, drop_mask = ctx.saved_tensors
        b, l = ctx.bl
        f = ctx.fused
        dqkv = ops.attn_small_bwd_fused(qkv, probs, do, b, l, ctx.n_heads, ctx.scale, drop_mask)
        wt = f["wt"] if all(w._version == v for w, v in zip(f["weights"], f["versions"])) else None
        if wt is not None:
            from .trunk import Conv3dP

            if Conv3dP._pending_arenas:  # an asynchronous refresh of the images may be in flight
                Conv3dP.join_pending_refresh()
        for w, g in zip(f["weights"], f["grads"]):  # re-attach if something replaced .grad
            if w.grad is None or w.grad.data_ptr() != g.data_ptr():
                w.grad = g
        dx, _, _ = ops.linear_bwd(dqkv, x2, f["w"], need_dx=True, has_bias=False, dw_out=f["dw"], wt=wt,
                                  dx_res=ctx.route.pop("dx", None) if ctx.route is not None else None)
        return dx.reshape(b, l, -1), None, None, None, None, None


class AddLayerNormFn(torch.autograd.Function):
    """LayerNorm(x + r * rmask) (vs_add_layernorm_*); rmask = residual-dropout mask or None."""

    @staticmethod
    def forward(ctx, x, r, gamma, beta, eps, rmask, route):
        ctx.route = route
        shp = x.shape
        x2, r2 = x.reshape(-1, shp[-1]).contiguous(), r.reshape(-1, shp[-1]).contiguous()
        rows, d = x2.shape
        ctx.lazy_ok = (_Lazy.enabled and x2.is_cuda and ops.ln_linear_ok(rows, d) and x2.dtype == torch.float32
                       and r2.dtype == torch.float32 and gamma.dtype == torch.float32
                       and all(t.data_ptr() % 16 == 0 for t in (x2, r2, gamma, beta))
                       and (rmask is None or rmask.data_ptr() % 16 == 0))
        if ctx.lazy_ok and _Lazy.active:
            _Lazy.flush_fwd()  # (an earlier parked one that no linear took)
            y = torch.empty_like(x2)
            mean = torch.empty(rows, dtype=torch.float32, device=x2.device)
            rstd = torch.empty(rows, dtype=torch.float32, device=x2.device)
            _Lazy.fwd = dict(x=x2, r=r2, gamma=gamma, beta=beta, eps=eps, rmask=rmask, y=y, mean=mean, rstd=rstd)
        else:
            _Lazy.flush_fwd()
            y, mean, rstd = ops.add_layernorm_fwd(x2, r2, gamma, beta, eps, rmask)
        ctx.save_for_backward(x2, r2, gamma, mean, rstd, rmask)
        ctx.shp = shp
        # arena parameters take their gradient in place (overwrite), like LinearFn
        direct = getattr(gamma, "_vs_direct_grad", False) and getattr(beta, "_vs_direct_grad", False)
        ctx.params = (gamma, beta) if direct else None
        return y.reshape(shp)

    @staticmethod
    def backward(ctx, dy):
        x2, r2, gamma, mean, rstd, rmask = ctx.saved_tensors
        direct = ctx.params is not None and ctx.params[0].grad is not None and ctx.params[1].grad is not None
        _Lazy.flush_bwd()
        dy2 = dy.reshape(x2.shape)
        if (ctx.lazy_ok and direct and ctx.route is not None and dy2.is_contiguous() and dy2.dtype == torch.float32
                and dy2.data_ptr() % 16 == 0):
            # parked: the backward of the linear that receives dr (the wrapped layer's last op) runs both.  Only with
            # in-place parameter gradients and the routed dx -- nothing autograd does before that launch may read
            # a tensor it fills.
            dx = torch.empty_like(x2)
            dr = torch.empty_like(x2) if rmask is not None else dx
            _Lazy.bwd = dict(dy=dy2, x=x2, r=r2, gamma=gamma, mean=mean, rstd=rstd, rmask=rmask, dx=dx, dr=dr,
                             dg=ctx.params[0].grad, db=ctx.params[1].grad)
            torch.autograd.Variable._execution_engine.queue_callback(_Lazy.flush_bwd)
            ctx.route["dx"] = dx
            return None, dr.reshape(ctx.shp), None, None, None, None, None
        dx, dr, dg, db = ops.add_layernorm_bwd(
            dy.reshape(x2.shape), x2, r2, gamma, mean, rstd, rmask,
            dg_out=ctx.params[0].grad if direct else None, db_out=ctx.params[1].grad if direct else None)
        gx = dx.reshape(ctx.shp)
        if ctx.route is not None:  # x's other consumer adds it to its own input gradient (no add launch)
            ctx.route["dx"] = dx
            gx = None
        if direct:
            return gx, dr.reshape(ctx.shp), None, None, None, None, None
        return gx, dr.reshape(ctx.shp), dg, db, None, None, None


_masks = ops.DropoutPool()  # one generator launch per encoder pass


class EncoderStackFn(torch.autograd.Function):
    """All `EncoderLayer`s of an `Encoder` on <= 8 token rows as TWO launches (forward, backward) instead of 7 + ~9 per
    layer: `ops.TxStack` / `vs_txenc_stack_run` walks the same kernels' bodies stage by stage behind grid barriers
    -- bitwise the per-op path (tests/test_gpu_txenc.py).  Off by default (VS_TXENC_STACK=1 switches it on): a grid
    barrier that is correct across the eight XCDs costs 9.4 us (2.9 us without its fences) against the 1.5 us of a
    kernel boundary, and one resident block per CU hides less latency than eight: the step is 0.9 ms slower with it.  Parameter gradients are written in place (arena
    parameters only, like LinearFn / AddLayerNormFn); the intermediate layer outputs it also returns are not
    differentiable."""

    @staticmethod
    def eligible(enc, x, mask):
        # opt-in: measured SLOWER than the per-op launches (profiles/r02_txenc_stack.txt) -- kept as the ablation
        if os.environ.get("VS_TXENC_STACK", "0") != "1" or mask is not None or enc.pe:
            return False
        if not (x.is_cuda and x.dim() == 3 and torch.is_grad_enabled() and x.requires_grad and len(enc.layers) > 0):
            return False
        b, l, d = x.shape
        if b * l > 8 or l > 16 or d % 4 or d > 2048:
            return False
        for layer in enc.layers:
            mh, ff = layer.selfattn.layer, layer.feedforward.layer
            f = mh._qkv
            if f is None or f["w"].data_ptr() != mh.wq.weight.data_ptr() or f["w"].shape != (3 * d, d):
                return False
            if not all(w._version == v for w, v in zip(f["weights"], f["versions"])):
                return False
            if d % mh.n_heads or ff.linear1.weight.shape[0] % 4 or ff.linear1.weight.shape[0] > 4096:
                return False
            for w in (mh.wo.weight, ff.linear1.weight, ff.linear2.weight):
                if getattr(w, "_vs_wt", None) is None or getattr(w, "_vs_wt_version", None) != w._version:
                    return False
            ps = [mh.wo.weight, ff.linear1.weight, ff.linear1.bias, ff.linear2.weight, ff.linear2.bias,
                  layer.selfattn.layernorm.weight, layer.selfattn.layernorm.bias,
                  layer.feedforward.layernorm.weight, layer.feedforward.layernorm.bias] + list(f["weights"])
            if not all(getattr(p, "_vs_direct_grad", False) and p.grad is not None for p in ps):
                return False
        return True

    @staticmethod
    def forward(ctx, x, enc):
        b, l, d = x.shape
        rows, dev = b * l, x.device
        x2 = x.reshape(rows, d)
        st = ops.TxStack(dev)
        saved = []
        f32 = dict(dtype=torch.float32, device=dev)
        xin = x2
        for layer in enc.layers:
            sa, fb = layer.selfattn, layer.feedforward
            mh, ff = sa.layer, fb.layer
            h_dim = ff.linear1.weight.shape[0]
            training = enc.training
            # the masks in the order the per-op path asks for them (same generator stream, same masks)
            pa, p1, p2 = mh.attention.dropout.p, sa.dropout.p, fb.dropout.p
            am = _masks.get((b, mh.n_heads, l, l), pa, dev) if training and pa > 0 else None
            m1 = _masks.get((rows, d), p1, dev) if training and p1 > 0 else None
            m2 = _masks.get((rows, d), p2, dev) if training and p2 > 0 else None
            buf = torch.empty(rows * (3 * d + 5 * d + h_dim) + b * mh.n_heads * l * l + 4 * rows, **f32)
            o = 0

            def take(n, shape):
                nonlocal o
                t = buf[o:o + n].view(shape)
                o += n
                return t
            qkv, att, br = take(rows * 3 * d, (rows, 3 * d)), take(rows * d, (rows, d)), take(rows * d, (rows, d))
            y1, hh = take(rows * d, (rows, d)), take(rows * h_dim, (rows, h_dim))
            ffo, y2 = take(rows * d, (rows, d)), take(rows * d, (rows, d))
            probs = take(b * mh.n_heads * l * l, (b, mh.n_heads, l, l))
            mean1, rstd1, mean2, rstd2 = (take(rows, (rows,)) for _ in range(4))
            fq = mh._qkv
            st.linear(xin, fq["w"], None, qkv)
            st.attn_fwd(qkv, att, probs, am, b, l, mh.n_heads, mh.attention.scale)
            st.linear(att, mh.wo.weight, None, br)
            st.add_layernorm(xin, br, m1, sa.layernorm.weight, sa.layernorm.bias, y1, mean1, rstd1, sa.layernorm.eps)
            st.linear(y1, ff.linear1.weight, ff.linear1.bias, hh, act=1)
            st.linear(hh, ff.linear2.weight, ff.linear2.bias, ffo)
            st.add_layernorm(y1, ffo, m2, fb.layernorm.weight, fb.layernorm.bias, y2, mean2, rstd2,
                             fb.layernorm.eps)
            saved.append(dict(xin=xin, qkv=qkv, att=att, br=br, y1=y1, hh=hh, ffo=ffo, y2=y2, probs=probs, am=am,
                              m1=m1, m2=m2, mean1=mean1, rstd1=rstd1, mean2=mean2, rstd2=rstd2))
            xin = y2
        st.run()
        ctx.enc, ctx.saved, ctx.bl, ctx.fwd_stack = enc, saved, (b, l, d), st
        outs = tuple(r["y2"].view(b, l, d) for r in saved)
        ctx.mark_non_differentiable(*outs[:-1])
        return outs

    @staticmethod
    def backward(ctx, *gouts):
        enc, saved = ctx.enc, ctx.saved
        b, l, d = ctx.bl
        rows = b * l
        g = gouts[-1].reshape(rows, d).contiguous()
        dev = g.device
        from .trunk import Conv3dP

        if Conv3dP._pending_arenas:  # the [K][N] weight images are refreshed on a side stream at the step's start
            Conv3dP.join_pending_refresh()
        st = ops.TxStack(dev)
        ga, gb = g, None
        for layer, r in zip(reversed(enc.layers), reversed(saved)):
            sa, fb = layer.selfattn, layer.feedforward
            mh, ff = sa.layer, fb.layer
            fq = mh._qkv
            for w, gr in zip(fq["weights"], fq["grads"]):  # re-attach if something replaced .grad
                if w.grad is None or w.grad.data_ptr() != gr.data_ptr():
                    w.grad = gr
            h_dim = ff.linear1.weight.shape[0]
            buf = torch.empty(rows * (7 * d + h_dim + 3 * d), dtype=torch.float32, device=dev)
            o = 0

            def take(n, shape):
                nonlocal o
                t = buf[o:o + n].view(shape)
                o += n
                return t
            d_y1a, d_f, d_y1b, d_xa, d_br, d_o, d_xb = (take(rows * d, (rows, d)) for _ in range(7))
            d_h, dqkv = take(rows * h_dim, (rows, h_dim)), take(rows * 3 * d, (rows, 3 * d))
            st.add_layernorm_bwd(ga, gb, r["y1"], r["ffo"], r["m2"], fb.layernorm.weight, r["mean2"], r["rstd2"],
                                 d_y1a, d_f, fb.layernorm.weight.grad, fb.layernorm.bias.grad)
            st.linear_bwd(d_f, None, r["hh"], ff.linear2.weight._vs_wt, d_h, ff.linear2.weight.grad,
                          ff.linear2.bias.grad)
            st.linear_bwd(d_h, r["hh"], r["y1"], ff.linear1.weight._vs_wt, d_y1b, ff.linear1.weight.grad,
                          ff.linear1.bias.grad)
            st.add_layernorm_bwd(d_y1a, d_y1b, r["xin"], r["br"], r["m1"], sa.layernorm.weight, r["mean1"],
                                 r["rstd1"], d_xa, d_br, sa.layernorm.weight.grad, sa.layernorm.bias.grad)
            st.linear_bwd(d_br, None, r["att"], mh.wo.weight._vs_wt, d_o, mh.wo.weight.grad, None)
            st.attn_bwd(r["qkv"], d_o, r["probs"], r["am"], dqkv, b, l, mh.n_heads, mh.attention.scale)
            st.linear_bwd(dqkv, None, r["xin"], fq["wt"], d_xb, fq["dw"], None)
            ga, gb = d_xa, d_xb
        dx = torch.empty((rows, d), dtype=torch.float32, device=dev)
        st.add(ga, gb, dx)
        st.run()
        ctx.saved = None
        return dx.view(b, l, d), None


def hip_linear(mod, x, relu=False, route=None):
    return LinearFn.apply(x, mod.weight, mod.bias, relu, route)


def _take_route(mod):
    """The gradient route a ResidualBlock offers the layer it wraps (`_route`), accepted: marked armed."""
    route = getattr(mod, "_route", None)
    if route is not None:
        route["armed"] = True
    return route


class Attention(nn.Module):
    def __init__(self, d_key, drop_ratio, causal):
        super().__init__()
        self.scale = math.sqrt(d_key)
        self.dropout = nn.Dropout(drop_ratio)
        self.causal = causal
        if causal:
            raise NotImplementedError("causal attention is not on the VidSitu hot path")


class MultiHead(nn.Module):
    def __init__(self, d_key, d_value, n_heads, drop_ratio, causal=False):
        super().__init__()
        self.attention = Attention(d_key, drop_ratio, causal=causal)
        self.wq = nn.Linear(d_key, d_key, bias=False)
        self.wk = nn.Linear(d_key, d_key, bias=False)
        self.wv = nn.Linear(d_value, d_value, bias=False)
        self.wo = nn.Linear(d_value, d_key, bias=False)
        self.n_heads = n_heads
        self._qkv = None  # set by ParamArena when wq / wk / wv sit back to back in the arena

    def forward(self, query, key, value):
        p = self.attention.dropout.p
        fused = self._qkv
        if fused is not None and query is key and key is value and query.dim() == 3 and query.is_cuda \
                and torch.is_grad_enabled() and query.requires_grad \
                and fused["w"].data_ptr() == self.wq.weight.data_ptr():
            b, l, _ = query.shape
            mask = _masks.get((b, self.n_heads, l, l), p, query.device) if self.training and p > 0 else None
            o = FusedQKVAttnFn.apply(query, fused, self.n_heads, self.attention.scale, mask, _take_route(self))
            return hip_linear(self.wo, o)
        q, k, v = hip_linear(self.wq, query), hip_linear(self.wk, key), hip_linear(self.wv, value)
        mask = None
        if self.training and p > 0:  # transformer_code.py:48 dropout(softmax(...))
            b, l, _ = q.shape
            mask = _masks.get((b, self.n_heads, l, l), p, q.device)
        o = AttnSmallFn.apply(q, k, v, self.n_heads, self.attention.scale, mask)
        return hip_linear(self.wo, o)


class FeedForward(nn.Module):
    def __init__(self, d_model, d_hidden):
        super().__init__()
        self.linear1 = nn.Linear(d_model, d_hidden)
        self.linear2 = nn.Linear(d_hidden, d_model)

    def forward(self, x):
        return hip_linear(self.linear2, hip_linear(self.linear1, x, relu=True, route=_take_route(self)))


class ResidualBlock(nn.Module):
    def __init__(self, layer, d_model, drop_ratio):
        super().__init__()
        self.layer = layer
        self.dropout = nn.Dropout(drop_ratio)
        self.layernorm = nn.LayerNorm(d_model)

    # x feeds the wrapped layer and the residual add: autograd would sum the two gradients of x with an add
    # launch (12 per step on the critical path of the 8-token section).  Instead the LayerNorm's backward hands its
    # dx to the layer's first op (a dict both hold), whose data-gradient kernel adds it in its epilogue
    # (`vs_linear_bwd_fused_res`): the same sum, rounded once.  VS_RESIDUAL_ROUTE=0 = autograd's add.
    route_grads = os.environ.get("VS_RESIDUAL_ROUTE", "1") != "0"

    def forward(self, *x):
        route = None
        if ResidualBlock.route_grads and torch.is_grad_enabled() and x[0].requires_grad and x[0].is_cuda \
                and all(t is x[0] for t in x):
            route = {"armed": False}
        self.layer._route = route
        try:
            branch = self.layer(*x)
        finally:
            self.layer._route = None
        if route is not None and not route["armed"]:
            route = None
        rmask = None
        if self.training and self.dropout.p > 0:  # transformer_code.py:30 x + dropout(layer(x))
            rmask = _masks.get((branch.numel() // branch.shape[-1], branch.shape[-1]),
                                self.dropout.p, branch.device)
        return AddLayerNormFn.apply(x[0], branch, self.layernorm.weight, self.layernorm.bias,
                                    self.layernorm.eps, rmask, route)


class EncoderLayer(nn.Module):
    def __init__(self, d_model, d_hidden, n_heads, drop_ratio):
        super().__init__()
        self.selfattn = ResidualBlock(MultiHead(d_model, d_model, n_heads, drop_ratio), d_model,
                                      drop_ratio)
        self.feedforward = ResidualBlock(FeedForward(d_model, d_hidden), d_model, drop_ratio)

    def forward(self, x):
        return self.feedforward(self.selfattn(x, x, x))


class Encoder(nn.Module):
    def __init__(self, d_model, d_hidden, n_vocab, n_layers, n_heads, drop_ratio, pe):
        super().__init__()
        self.layers = nn.ModuleList(
            [EncoderLayer(d_model, d_hidden, n_heads, drop_ratio) for _ in range(n_layers)])
        self.dropout = nn.Dropout(drop_ratio)
        self.pe = pe

    def forward(self, x, mask=None):
        if self.pe:
            raise NotImplementedError
        if not x.is_cuda:
            raise ops._lib.VsError("the TxEncoder runs on the HIP kernels only (GPU tensor required)")
        x = x.float().contiguous()
        if self.training:
            _masks.begin(x.device)
        if mask is not None:
            x = x * mask
        encoding = [x]
        if EncoderStackFn.eligible(self, x, mask):
            return encoding + list(EncoderStackFn.apply(x, self))
        prev = _Lazy.active
        _Lazy.active = _Lazy.enabled and mask is None and torch.is_grad_enabled()
        try:
            for layer in self.layers:
                x = layer(x)
                if mask is not None:
                    x = x * mask
                encoding.append(x)
        finally:
            _Lazy.active = prev
            _Lazy.flush_fwd()  # the last LayerNorm has no linear of this encoder behind it
        return encoding


class Transformer(nn.Module):
    def __init__(self, d_model, n_vocab_src, vocab_trg, d_hidden=2048, n_layers=6, n_heads=8,
                 drop_ratio=0.1, pe=False):
        super().__init__()
        self.encoder = Encoder(d_model, d_hidden, n_vocab_src, n_layers, n_heads, drop_ratio, pe)

    def forward(self, x):
        return self.encoder(x)[-1]

    def all_outputs(self, x):
        return self.encoder(x)
