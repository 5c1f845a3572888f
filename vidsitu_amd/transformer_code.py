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


class LinearFn(torch.autograd.Function):
    """y = act(x @ W^T + b) on vs_linear_*  (x: [..., K] fp32)."""

    @staticmethod
    def forward(ctx, x, w, b, relu, route):
        ctx.route = route  # see ResidualBlock: the residual path's gradient joins this layer's dx in the kernel
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
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
        o, probs = ops.attn_small_fwd(q, k, v, n_heads, scale, drop_mask)
        ctx.save_for_backward(q, k, v, probs, drop_mask)
        ctx.n_heads, ctx.scale = n_heads, scale
        return o

    @staticmethod
    def backward(ctx, do):
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
        qkv = ops.linear_fwd(x2, fused["w"], None, False)
        o, probs = ops.attn_small_fwd_fused(qkv, b, l, n_heads, scale, drop_mask)
        ctx.save_for_backward(x2, qkv, probs, drop_mask)
        ctx.fused, ctx.n_heads, ctx.scale, ctx.bl = fused, n_heads, scale, (b, l)
        return o

    @staticmethod
    def backward(ctx, do):
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
        for layer in self.layers:
            x = layer(x)
            if mask is not None:
                x = x * mask
            encoding.append(x)
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
