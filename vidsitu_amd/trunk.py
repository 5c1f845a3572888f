"""SlowFast / ResNet video trunk on the HIP kernels.

Host-side mirror of the module tree the reference reaches through
`slowfast.models.video_model_builder.{SlowFast,ResNet}` and subclasses in
`vidsitu_code/mdl_sf_base.py:20-62` (`SlowFast_FeatModel`, `ResNet_FeatModel`):
same attribute names (`s1, s1_fuse, s2, ... s5, pathway{p}_pool`), same
state_dict keys, same `forward_features(list) -> list`.

Compute never touches torch.nn.functional: every layer is a launch of
libvidsitu_hip.so through `ops`.  Activations are bf16 channels-last; the
tensors handed back keep the reference's logical NCDHW shape.

  eval : conv epilogue applies the folded BN (+residual, +ReLU)   -> 1 launch / conv
  train: conv emits raw bf16 + fp32 batch-stat partials -> finalize -> apply
         (+residual, +ReLU); the backward is hand written (BN reduce/apply,
         conv dgrad / wgrad) and writes parameter gradients straight into
         `param.grad` (one fp32 arena), so DDP-style averaging is one all-reduce.
"""
import contextlib
import os

import torch
from torch import nn

from . import ops

STAGE_DEPTH = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), "tiny": (1, 0, 0, 1), "mini": (1, 1, 1, 1)}
TEMPORAL_KERNEL_BASIS = {
    "c2d": [[[1]], [[1]], [[1]], [[1]], [[1]]],
    "i3d": [[[5]], [[3]], [[3, 1]], [[3, 1]], [[1, 3]]],
    "slow": [[[1]], [[1]], [[1]], [[3]], [[3]]],
    "slowfast": [[[1], [5]], [[1], [3]], [[1], [3]], [[3], [3]], [[3], [3]]],
}
POOL1 = {
    "c2d": [[2, 1, 1]],
    "i3d": [[2, 1, 1]],
    "slow": [[1, 1, 1]],
    "slowfast": [[1, 1, 1], [1, 1, 1]],
}


# ----------------------------------------------------------------------------
# parameter holders (names = upstream names)
# ----------------------------------------------------------------------------
class Conv3dP(nn.Module):
    """Conv3d parameters (bias-free except in the non-local block).  `weight` is the fp32 master in the reference's
    logical shape [Cout,Cin,kT,kH,kW] with channels-last memory."""

    def __init__(self, cin, cout, k, s=(1, 1, 1), p=(0, 0, 0), bias=False):
        super().__init__()
        self.cin, self.cout = cin, cout
        # only the non-local block's 1x1x1 convs carry a bias (slowfast nonlocal_helper: nn.Conv3d default)
        self.bias = nn.Parameter(torch.zeros(cout)) if bias else None
        self.k, self.s, self.p = tuple(k), tuple(s), tuple(p)
        self.cin_pad = (cin + 7) // 8 * 8
        w = torch.empty(cout, *self.k, cin).permute(0, 4, 1, 2, 3)
        fan_out = cout * k[0] * k[1] * k[2]
        w.normal_(0.0, (2.0 / fan_out) ** 0.5)  # c2_msra_fill
        self.weight = nn.Parameter(w)
        self.w_bf16 = None  # [Cout, cin_pad, k] channels-last, refreshed by the trunk
        self.wt_bf16 = None  # transposed image for dgrad
        # the two stems (Cin = 3, [kT,7,7], stride [1,2,2]) run on the dedicated kernel
        cp = (cout + 15) // 16 * 16
        stem_lds = cp * (self.k[0] * 7 * 64 + 16) + self.k[0] * 21 * 320 + 128 * cp * 4 + 32 * cp
        self.is_stem = (cin == 3 and self.k[1:] == (7, 7) and self.s == (1, 2, 2)
                        and self.p == (self.k[0] // 2, 3, 3) and cout <= 64
                        and stem_lds <= 150 * 1024)  # whole packed weight must sit in LDS
        self.w_stem = None

    def refresh(self):
        """Rebuild the bf16 kernel-layout copy (and, when its storage is managed by a
        ParamArena, the transposed dgrad image) of this conv's weight."""
        w = self.weight.detach()
        if self.w_bf16 is None or self.w_bf16.device != w.device:
            self.w_bf16 = torch.zeros(
                (self.cout, *self.k, self.cin_pad), dtype=ops.BF16, device=w.device
            ).permute(0, 4, 1, 2, 3)
            self.arena_managed = False
        if self.cin_pad == self.cin:
            ops.cast_bf16(w, self.w_bf16)  # identical memory order
        else:
            self.w_bf16[:, : self.cin].copy_(w)
        if self.is_stem:
            self.w_stem = ops.pack_stem_weight(w, out=self.w_stem if self.w_stem is not None and
                                               self.w_stem.device == w.device else None)
        if getattr(self, "arena_managed", False):
            ops.weight_transpose(self.w_bf16, out=self.wt_bf16)
        else:
            self.wt_bf16 = None

    # ParamArenas whose asynchronous refresh of the dgrad images is in flight (one entry per arena: two
    # models in one process each join their own side stream)
    _pending_arenas = set()

    @staticmethod
    def join_pending_refresh():
        for a in list(Conv3dP._pending_arenas):
            a._join_transposes()

    def lo(self):
        """Eval-mode split weights (`_Unit.split_weights`): the bf16 image of what rounding the fp32 master to bf16
        dropped, W_lo = bf16(W - float(bf16(W))), in the kernels' layout (and packed for the stem kernel); rebuilt
        when the master's version counter moved."""
        w = self.weight.detach()
        key = (w._version, w.data_ptr())
        if getattr(self, "_lo_key", None) != key:
            rest = w - w.to(ops.BF16).float()
            lo = torch.zeros((self.cout, *self.k, self.cin_pad), dtype=ops.BF16, device=w.device).permute(0, 4, 1, 2, 3)
            lo[:, : self.cin].copy_(rest)
            self._lo = lo
            self._lo_stem = ops.pack_stem_weight(rest) if self.is_stem else None
            self._lo_key = key
        return self._lo_stem if self.is_stem else self._lo

    def wt(self):
        if Conv3dP._pending_arenas:
            Conv3dP.join_pending_refresh()
        if self.wt_bf16 is None:
            self.wt_bf16 = ops.weight_transpose(self.w_bf16)
        return self.wt_bf16


class BN3dP(nn.Module):
    def __init__(self, c, eps=1e-5, momentum=0.1, zero_init=False):
        super().__init__()
        self.num_features, self.eps, self.momentum = c, eps, momentum
        self.weight = nn.Parameter(torch.zeros(c) if zero_init else torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self.fold = None  # eval-mode (scale, shift), cached by the trunk
        # eval only: per-channel mean of conv(x, bf16(W) - W) of the convolution in front of this BN, measured by
        # VideoTrunk.calibrate_weight_rounding; the fold subtracts scale * wround_bias from the shift.  None: no correction.
        self.wround_bias = None


def _set_grad(param, g):
    """Write a gradient produced by a HIP kernel (overwrite semantics)."""
    if param.grad is None:
        param.grad = g if g.shape == param.shape else g.reshape(param.shape)
    else:
        param.grad.copy_(g)


class _WgradLanes:
    """Weight gradients have no consumer before the optimizer, so a conv_wgrad launch issued from the
    caller's (slow-pathway) stream goes to a side lane and runs beside the same unit's dgrad; the
    lane is joined right after that dgrad (`join_all`), operands kept alive until then.
    Measured alternatives (batch-8 train step, hipGraph; 17.24 ms without lanes, 16.71 ms with):
    joining only at the end of a stage 19.1 ms, joining one unit later 17.7 ms (wgrad then competes
    with the BN passes for bandwidth), lanes for the fast pathway as well 18.8 ms; a lane forked from
    AND joined back into the trunk's (already forked) side stream makes hipStreamEndCapture
    segfault on ROCm 7.2, so launches issued from that stream stay inline.  VS_WGRAD_LANES=0: off."""

    enabled = os.environ.get("VS_WGRAD_LANES", "1") != "0"
    # units the lane may lag behind the issuing stream before that stream waits for it (0: joined right after
    # the unit's dgrad).  VS_WGRAD_LAG: A/B knob.
    lag = int(os.environ.get("VS_WGRAD_LAG", "0"))
    # device index -> [issuing stream, lane stream, queue of (event, keepalive) per launched unit].  Lane streams
    # are created once (never inside a hipGraph capture, where creating a stream is an unsafe call).
    lanes = {}

    # A fork + join pair costs ~17 us in a replayed hipGraph on this system (tools/graph_edge_cost.py): a weight
    # gradient only goes to the lane when its convolution is at least this large (GFLOP of the unit's GEMM);
    # smaller ones run inline in front of the dgrad.  VS_WGRAD_LANE_MIN_GFLOP: sweep knob.
    min_gflop = float(os.environ.get("VS_WGRAD_LANE_MIN_GFLOP", "0"))
    defer_active = False  # inside VideoTrunk._backward_segment with ops.REDUCE_MERGE (see there)

    @classmethod
    def run(cls, fn, *keep, gflop=None):
        """fn runs on the lane; if it returns a callable (the slab reduce of ops.conv_wgrad_split) that runs on the
        lane too, but BEHIND the event the issuing stream waits for: the reduce reads no operand of the unit, so the
        issuing stream goes on as soon as the wgrad kernel itself is done (~7 us per unit off its critical path)."""
        if not cls.enabled or not keep[0].is_cuda or (gflop is not None and gflop < cls.min_gflop):
            tail = fn()
            return tail() if callable(tail) else None
        main = torch.cuda.current_stream()
        dev = main.device.index
        side = VideoTrunk._side_streams.get(dev)
        if side is not None and side.cuda_stream == main.cuda_stream:
            tail = fn()  # issued from the fast pathway's stream: no nested fork
            return tail() if callable(tail) else None
        lane = cls.lanes.get(dev)
        if lane is None:
            lane = cls.lanes[dev] = [main, torch.cuda.Stream(device=main.device), [], False]
        if lane[2] and lane[0].cuda_stream != main.cuda_stream:  # never leave a lane un-joined
            cls._drain(lane, 0)
        lane[0] = main
        lane[1].wait_stream(main)
        deferring = _WgradLanes.defer_active
        if deferring:
            ops.wgrad_reduce_defer(2)  # the lane's completion event must cover the slab reduce: launched as usual
        with torch.cuda.stream(lane[1]):
            tail = fn()
            ev = torch.cuda.Event()
            ev.record(lane[1])
            if callable(tail):
                tail()
                lane[3] = True  # work behind the last event: a full join waits for the stream, not the event
        if deferring:
            ops.wgrad_reduce_defer(1)
        lane[2].append((ev, keep))

    @staticmethod
    def _drain(lane, lag):
        while len(lane[2]) > lag:
            ev, _ = lane[2].pop(0)
            lane[0].wait_event(ev)

    @classmethod
    def join_unit(cls):
        """After a unit's dgrad: wait for the weight gradients launched more than `lag` units ago."""
        for lane in cls.lanes.values():
            cls._drain(lane, cls.lag)

    @classmethod
    def join_all(cls):
        for lane in cls.lanes.values():
            cls._drain(lane, 0)
            if lane[3]:
                lane[0].wait_stream(lane[1])
                lane[3] = False


class _Unit:
    """conv -> BN (-> +residual) (-> ReLU) executed on the HIP kernels."""

    trace = None  # debugging: set to a list to record (conv, y, z, mean, invstd) per unit
    wgrad_batch = None  # the running trunk backward's ops.WgradBatch (set by VideoTrunk._backward_segment)
    # VS_WGRAD_TAIL=1: the slab reduce of a weight gradient runs behind the event its unit's stream waits for (the
    # reduce reads no operand of the unit).  Measured: 16.3 ms vs 13.5 ms per step under hipGraph replay -- like the
    # lagged join (`_WgradLanes.lag`), ANY lane work that is still running when the issuing stream starts the next
    # unit's BN kernels costs ~25 us per unit; the replay then takes as long as the one-stream graph (17.0 ms).
    # Eager (host-bound, 24-26 ms) does not show it.  Off.
    split_wgrad_reduce = os.environ.get("VS_WGRAD_TAIL", "0") == "1"
    # BN-backward sums of the a / b units of a bottleneck emitted by the consuming convolution's dgrad
    # epilogue (vs_conv_dgrad_bnstats) instead of a reduce pass of their own; VS_FUSE_BN_SUMS=0 = A/B switch
    fuse_bn_sums = os.environ.get("VS_FUSE_BN_SUMS", "1") != "0"
    fuse_sc_sums = os.environ.get("VS_FUSE_SC_SUMS", "1") != "0"  # the shortcut unit's sums from the same epilogue
    # data gradient + weight gradient of a unit as ONE launch where both run on the 128 x 128 ring kernels
    # (ops.conv_pair): no side lane, hence no fork / join edge pair (~17 us in a replayed graph).  VS_CONV_PAIR=0: lanes.
    pair_launch = os.environ.get("VS_CONV_PAIR", "1") != "0"
    # Eval only: split bf16 weights W = W_hi + W_lo, two launches per convolution (north_star's "logits within 1e-3 of
    # the reference": met in this mode, at about half the clips/s).  VS_EVAL_SPLIT_WEIGHTS=1.
    split_weights = os.environ.get("VS_EVAL_SPLIT_WEIGHTS", "0") == "1"
    _zeros = {}

    # VideoTrunk.calibrate_weight_rounding: a dict while the calibration pass runs (conv -> measured input means)
    calib = None

    @staticmethod
    def _calibrate(conv, bn, x):
        """Calibration pass (eval): the rounding of this convolution's fp32 weights to bf16 adds conv(x, dW),
        dW = bf16(W) - W, to its output.  The trimmed head averages the feature map over positions, so what reaches the
        logits is the position MEAN of that error -- per output channel sum_{taps, ci} dW[co, ci, tap] * mean(x[ci])
        (borders ignored) -- and a per-channel constant is something the folded BatchNorm shift can absorb: the fold
        subtracts scale * wround_bias.  x is the input this launch is about to read (its producers already corrected),
        so the means are those of the corrected path.  Host-side bookkeeping on [Cout, Cin] tensors; the activation
        means come from vs_colsum_bf16."""
        with torch.no_grad():
            w = conv.weight.detach()
            if conv.is_stem:
                # (round 6) the stems too: decoded frames are not zero-mean per video the way N(0,1) noise is -- a video's
                # brightness puts a channel mean of up to +-1 on the normalised input.  x = the packed C = 4 input
                x = x[0]
                mu = (ops.colsum_bf16(x) / float(ops.act_rows(x)))[: conv.cin]
                dw = (w.to(torch.bfloat16).float() - w).sum(dim=(2, 3, 4))  # [Cout, 3]
            else:
                mu = ops.colsum_bf16(x) / float(ops.act_rows(x))  # [Cin]
                dw = (conv.w_bf16[:, : conv.cin].float() - w).sum(dim=(2, 3, 4))  # [Cout, Cin]
            bn.wround_bias = (dw * mu.view(1, -1)).sum(dim=1)
            sc, sh = bn.fold_raw
            bn.fold = (sc, sh - sc * bn.wround_bias)
            _Unit.calib[conv] = mu

    @staticmethod
    def fwd(conv, bn, x, relu, residual=None, out=None, train=False, saved=None, pool=False, no_apply=False,
            x_affine=None, defer_apply=False, residual_affine=None):
        """pool (train mode, the stems): the unit's output goes through the [1,3,3] max-pool and nowhere else --
        BN + ReLU + pool run as one pass, the record gets the argmax bytes (`pool_idx`) and no `z`.
        Apply on load (train mode, `ResBlock.aol`): `no_apply` -- the unit stops behind its finalize and returns
        (y, (scale, shift)); its consumer is called with `x` = that raw y and `x_affine` = the constants and forms
        relu(y * scale + shift) on its operand fragments (ops.conv_fwd_aol / conv_wgrad_aol): the activation is never
        stored.
        Shortcut apply in the block's last pass (train mode, `ResBlock.fuse_sc_apply`): the shortcut unit is called with
        `defer_apply` -- it stops behind its finalize and returns (y, (scale, shift)) -- and the c unit with
        `residual_affine` = that pair: its apply pass forms the shortcut's normalised output on the fly (ops.bn_apply2,
        bitwise the two passes)."""
        if not train:
            if _Unit.calib is not None and (conv.is_stem or conv.cin_pad == conv.cin):
                _Unit._calibrate(conv, bn, x)
            scale, shift = bn.fold
            if conv.bias is not None:  # BN(conv + b) folded: the bias joins the shift
                shift = shift + conv.bias.detach() * scale
            if _Unit.split_weights:
                # Every convolution twice, the second time on what bf16 dropped of the fp32 weights.  Rounding the
                # WEIGHTS is a coherent error -- the same at every position, it survives the average pool -- and is
                # the whole distance between this path and the fp32 reference at the logits (3.0e-3 of 3.0e-3 on a
                # 224^2 SlowFast-R50 clip; the activations' rounding contributes 4e-4: tests/test_gpu_parity_full.py).
                # t = scale * (x * W_lo) (+ residual), then the usual launch with t as its residual.
                zero = _Unit._zeros.get((shift.numel(), str(shift.device)))
                if zero is None:
                    zero = _Unit._zeros[(shift.numel(), str(shift.device))] = torch.zeros_like(shift)
                if conv.is_stem:
                    t, _ = ops.stem_conv_fwd(x[0], conv.lo(), conv.cout, conv.k[0], scale=scale, shift=zero)
                    y, _ = ops.stem_conv_fwd(x[0], conv.w_stem, conv.cout, conv.k[0])
                    return ops.bn_apply(y, scale, shift, t, relu, out=out)
                t, _ = ops.conv_fwd(x, conv.lo(), conv.k, conv.s, conv.p, scale=scale, shift=zero, residual=residual)
                y, _ = ops.conv_fwd(x, conv.w_bf16, conv.k, conv.s, conv.p, out=out, scale=scale, shift=shift,
                                    residual=t, relu=relu)
                return y
            if conv.is_stem:
                y, _ = ops.stem_conv_fwd(x[0], conv.w_stem, conv.cout, conv.k[0], out=out,
                                         scale=scale, shift=shift, relu=relu)
                return y
            y, _ = ops.conv_fwd(x, conv.w_bf16, conv.k, conv.s, conv.p, out=out, scale=scale,
                                shift=shift, residual=residual, relu=relu)
            return y
        if conv.is_stem:
            y, partials = ops.stem_conv_fwd(x[0], conv.w_stem, conv.cout, conv.k[0], stats=True)
            x = x[0]  # the stem wgrad kernel reads the same C=4 packed input
        elif x_affine is not None:
            y, partials = ops.conv_fwd_aol(x, conv.w_bf16, x_affine[0], x_affine[1], stats=True)
        else:
            y, partials = ops.conv_fwd(x, conv.w_bf16, conv.k, conv.s, conv.p, stats=True)
        scale, shift, mean, invstd = ops.bn_finalize(
            partials, ops.act_rows(y), bn.weight, bn.bias, bn.running_mean, bn.running_var,
            bn.momentum, bn.eps, train=True)
        if conv.bias is not None:
            # train-mode BN removes a per-channel constant exactly, so the conv ran without its bias;
            # only the running mean sees it: mean(conv + b) = mean(conv) + b
            with torch.no_grad():
                bn.running_mean.add_(conv.bias.detach() * bn.momentum)
        if defer_apply:  # a ReLU-less unit whose output is formed by its only reader (the block's last apply pass)
            assert not relu and residual is None and not pool and x_affine is None
            saved.append(dict(conv=conv, bn=bn, x=x, y=y, z=None, zbits=None, mean=mean, invstd=invstd,
                              relu=False, has_res=False, x_affine=None))
            return y, (scale, shift)
        if no_apply:  # the consumer applies scale / shift / ReLU on load; backward recomputes the mask from y
            assert relu and residual is None and not pool
            saved.append(dict(conv=conv, bn=bn, x=x, y=y, z=None, zbits=None, mean=mean, invstd=invstd,
                              relu=True, has_res=False, aol=True, x_affine=x_affine))
            return y, (scale, shift)
        # a unit with a residual input cannot recompute its ReLU mask from y alone: keep it as bits
        if pool and not ops.bn_apply_maxpool_ok(y):
            pool, out = False, None  # the caller pools z itself (`out` was meant for the pooled tensor)
        if pool:
            pooled, pidx = ops.bn_apply_maxpool(y, scale, shift, out=out)
            saved.append(dict(conv=conv, bn=bn, x=x, y=y, z=None, zbits=None, mean=mean, invstd=invstd,
                              relu=True, has_res=False, pool_idx=pidx))
            return pooled
        if residual_affine is not None:
            y2, (sc2, sh2) = residual_affine
            assert relu and residual is None
            z, zbits = ops.bn_apply2(y, scale, shift, y2, sc2, sh2, out=out, want_bits=True)
            saved.append(dict(conv=conv, bn=bn, x=x, y=y, z=z, zbits=zbits, mean=mean, invstd=invstd,
                              relu=True, has_res=True, x_affine=x_affine))
            return z
        want_bits = relu and residual is not None and y.shape[1] % 8 == 0 and \
            ((y.shape[1] // 8) & (y.shape[1] // 8 - 1)) == 0
        if want_bits:
            z, zbits = ops.bn_apply(y, scale, shift, residual, relu, out=out, want_bits=True)
        else:
            z, zbits = ops.bn_apply(y, scale, shift, residual, relu, out=out), None
        if _Unit.trace is not None:
            _Unit.trace.append((conv, y.float().cpu(), z.float().cpu(), mean.cpu(), invstd.cpu()))
        saved.append(dict(conv=conv, bn=bn, x=x, y=y, z=z, zbits=zbits, mean=mean, invstd=invstd,
                          relu=relu, has_res=residual is not None, x_affine=x_affine))
        return z

    @staticmethod
    def bwd(rec, dz, need_dx=True, want_dres=False, dx_residual=None, masked=False, producer=None,
            dz_bits=None, dx_residual_bits=None, inplace=False, dy_ready=None, wgrad_sink=None):
        """Returns (dx|None, dres|None).  `masked`: dz already carries the ReLU mask.
        `producer`: the record of the unit whose output is this unit's only input (the a -> b and b -> c
        links of a bottleneck): this unit's dgrad then also emits the producer's BN-backward sums
        (`_Unit.fuse_bn_sums`), which spares the producer's reduce pass over dz and y.
        `dz_bits`: dz is an UNMASKED gradient and these the ReLU bits to apply to it (the shortcut unit of a
        ResBlock reads the block's output gradient through the block's own mask instead of a masked copy).
        `dx_residual_bits`: the same for the gradient added in the dgrad epilogue.
        `inplace`: the dgrad accumulates into dx_residual.
        `dy_ready`: the unit's BN backward already ran (`ResBlock.bwd`: the c and shortcut units in one pass) and this
        is its dy -- only the convolution's gradients are left.
        `wgrad_sink`: a list -- the unit's weight gradient is not launched here but appended as (dy, x, conv) for the
        block's grouped launch (`ResBlock.bwd`, ops.conv_wgrad_group)."""
        conv, bn = rec["conv"], rec["bn"]
        if dy_ready is not None:
            assert not want_dres
            x = rec["x"]
            pair = (_Unit.pair_launch and need_dx and not conv.is_stem and conv.cin_pad == conv.cin and dy_ready.is_cuda
                    and _Unit.wgrad_batch is None and not _Unit.split_wgrad_reduce and wgrad_sink is None)
            pair_ctx = ops.conv_pair() if pair else contextlib.nullcontext()
            pair_ctx.__enter__()
            try:
                dx = _Unit._bwd_convs(conv, dy_ready, x, need_dx, producer, dx_residual, dx_residual_bits, inplace, pair,
                                      rec.get("x_affine"), wgrad_sink)
            finally:
                pair_ctx.__exit__(None, None, None)
            _WgradLanes.join_unit()
            return dx, None
        relu = rec["relu"] and not masked
        if bn.weight.grad is None:
            bn.weight.grad = torch.zeros_like(bn.weight)
        if bn.bias.grad is None:
            bn.bias.grad = torch.zeros_like(bn.bias)
        # without a residual input the ReLU mask is recomputed from y (z is not read)
        zbits = rec.get("zbits") if relu else None
        zmask = rec["z"] if (relu and rec["has_res"] and zbits is None) else None
        if dz_bits is not None:
            assert not rec["relu"], "dz_bits belongs to a unit without a ReLU of its own"
            relu, zbits, zmask = True, dz_bits, None
        pidx = rec.get("pool_idx")  # a stem unit run with pool=True: dz is the POOLED tensor's gradient
        dy, dres, _, _ = ops.bn_bwd(
            None if pidx is not None else dz, zmask, rec["y"], rec["mean"], rec["invstd"], bn.weight, relu, want_dres,
            dgamma=bn.weight.grad, dbeta=bn.bias.grad, beta=bn.bias, zbits=zbits,
            partial=rec.pop("bwd_partial", None), pool_src=(dz, pidx) if pidx is not None else None)
        x = rec["x"]
        if conv.bias is not None:  # analytically zero behind a train-mode BN
            if conv.bias.grad is None:
                conv.bias.grad = torch.zeros_like(conv.bias)
            else:
                conv.bias.grad.zero_()
        # one launch for the unit's two gradients where both kernels allow it (`pair_launch`): the weight gradient is
        # then issued inline (recorded, like the data gradient below) instead of on the lane
        pair = (_Unit.pair_launch and need_dx and not conv.is_stem and conv.cin_pad == conv.cin and dy.is_cuda
                and _Unit.wgrad_batch is None and not _Unit.split_wgrad_reduce and wgrad_sink is None)
        pair_ctx = ops.conv_pair() if pair else contextlib.nullcontext()
        pair_ctx.__enter__()
        try:
            dx = _Unit._bwd_convs(conv, dy, x, need_dx, producer, dx_residual, dx_residual_bits, inplace, pair,
                                  rec.get("x_affine"), wgrad_sink)
        finally:
            pair_ctx.__exit__(None, None, None)
        _WgradLanes.join_unit()  # wgrad || dgrad of this unit (and, with a lag, of the next units)
        return dx, dres

    @staticmethod
    def _bwd_convs(conv, dy, x, need_dx, producer, dx_residual, dx_residual_bits, inplace, pair, x_affine=None,
                   wgrad_sink=None):
        """The unit's weight gradient (side lane, or inline when `pair`) and data gradient.  x_affine: x is the
        producer's raw output, the weight gradient applies its BN + ReLU on load.  wgrad_sink: see `bwd`."""
        if wgrad_sink is not None:
            if conv.weight.grad is None:
                conv.weight.grad = torch.empty_like(conv.weight)
            wgrad_sink.append((dy, x, conv))
        elif conv.is_stem:
            _WgradLanes.run(lambda: _set_grad(conv.weight, ops.stem_conv_wgrad(dy, x, conv.k[0])) and None, dy, x)
        elif conv.cin_pad == conv.cin:
            if conv.weight.grad is None:
                conv.weight.grad = torch.empty_like(conv.weight)
            # gradients that live in a ParamArena keep their address: their position-split partials go to the
            # trunk's WgradBatch and are summed by one launch per backward segment (VideoTrunk._flush_wgrads)
            batch = _Unit.wgrad_batch if getattr(conv.weight, "_vs_direct_grad", False) else None
            gf = 2e-9 * dy.numel() * conv.cin * conv.k[0] * conv.k[1] * conv.k[2]  # the unit's GEMM, GFLOP
            if pair:
                gf = -1.0  # inline
            if x_affine is not None:
                _WgradLanes.run(lambda: (ops.conv_wgrad_aol(dy, x, x_affine[0], x_affine[1], out=conv.weight.grad),
                                         None)[1], dy, x, gflop=gf)
            elif batch is not None:
                _WgradLanes.run(lambda: (ops.conv_wgrad(dy, x, conv.k, conv.s, conv.p, out=conv.weight.grad,
                                                        batch=batch), None)[1], dy, x, gflop=gf)
            elif _Unit.split_wgrad_reduce:
                _WgradLanes.run(lambda: ops.conv_wgrad_split(dy, x, conv.k, conv.s, conv.p, conv.weight.grad), dy, x,
                                gflop=gf)
            else:
                _WgradLanes.run(lambda: (ops.conv_wgrad(dy, x, conv.k, conv.s, conv.p, out=conv.weight.grad),
                                         None)[1], dy, x, gflop=gf)
        else:
            def legacy():
                dwp = ops.conv_wgrad(dy, x, conv.k, conv.s, conv.p)
                _set_grad(conv.weight, dwp[:, : conv.cin])
            _WgradLanes.run(legacy, dy, x)
        dx = None
        if need_dx:
            # the producer's complete dz is this dx when its output feeds this convolution and -- with
            # dx_residual -- the branch whose gradient arrives as that residual, and nothing else
            fuse = (_Unit.fuse_bn_sums and producer is not None and "conv" in producer and producer["relu"]
                    and (producer["y"] is x if producer.get("aol") else producer["z"] is x))
            if fuse and dx_residual is None:
                fuse = not producer["has_res"] and producer.get("zbits") is None
            elif fuse:
                fuse = producer.get("zbits") is not None
            part = None
            if fuse:
                pbn = producer["bn"]
                # the producer's block may have a shortcut unit: it receives the same masked gradient (one more
                # sum(g * xhat) from the same epilogue instead of a reduce pass of its own over dz and its y)
                sc = producer.get("sc_rec") if (_Unit.fuse_sc_sums and dx_residual is not None) else None
                res = ops.conv_dgrad(dy, conv.wt(), tuple(x.shape), conv.k, conv.s, conv.p,
                                     residual=dx_residual, residual_bits=dx_residual_bits,
                                     bn_stats=(producer["y"], producer["mean"], producer["invstd"],
                                               pbn.weight, pbn.bias, producer.get("zbits")),
                                     bn_stats2=(sc["y"], sc["mean"], sc["invstd"]) if sc is not None else None)
                dx, part = res[0], res[1]
                if part is not None:
                    producer["bwd_partial"] = part
                    if sc is not None and res[2] is not None:
                        sc["bwd_partial"] = res[2]
            else:
                dx = ops.conv_dgrad(dy, conv.wt(), tuple(x.shape), conv.k, conv.s, conv.p,
                                    residual=dx_residual, residual_bits=dx_residual_bits, inplace=inplace)
        return dx


class ResNetBasicStem(nn.Module):
    def __init__(self, cin, cout, kt, eps, mom):
        super().__init__()
        self.conv = Conv3dP(cin, cout, (kt, 7, 7), (1, 2, 2), (kt // 2, 3, 3))
        self.bn = BN3dP(cout, eps, mom)

    # train mode: BN + ReLU + pool in one pass (ops.bn_apply_maxpool): the full-resolution normalised tensor -- 154 MB for
    # the two stems of SlowFast-R50 at 8 clips -- is neither written nor re-read.  VS_STEM_POOL_FUSE=0: separate launches.
    # (The pool's backward inside the BN-backward passes, ops.bn_bwd(pool_src=...), exists as well.)
    fuse_pool = os.environ.get("VS_STEM_POOL_FUSE", "1") != "0"
    # the backward half is off by default: measured alone (tools/stem_pool_time.py) the forward pass is 62 vs 80 us
    # (slow stem) and 26 vs 42 us (fast stem), but gathering the pool's gradient inside BOTH backward passes costs
    # more instructions than the dense tensor costs bytes: 202 vs 149 us and 126 vs 83 us.  VS_STEM_POOL_FUSE_BWD=1.
    fuse_pool_bwd = os.environ.get("VS_STEM_POOL_FUSE_BWD", "0") == "1"

    def fwd(self, x, out, train, saved):
        if train and ResNetBasicStem.fuse_pool and _Unit.trace is None:
            z = _Unit.fwd(self.conv, self.bn, x, True, out=out, train=True, saved=saved, pool=True)
            if saved[-1].get("pool_idx") is not None:  # fused: z is the pooled tensor
                saved.append(dict(fused_pool=True))
                return z
        else:
            z = _Unit.fwd(self.conv, self.bn, x, True, train=train, saved=saved)
        y, idx = ops.maxpool_hw(z, out=out, want_idx=train)
        if train:
            saved.append(dict(pool_idx=idx, pool_in=tuple(z.shape)))
        return y

    def bwd(self, saved, dy):
        rec = saved.pop()
        if rec.get("fused_pool"):
            unit = saved.pop()
            if not ResNetBasicStem.fuse_pool_bwd:
                dy = ops.maxpool_hw_bwd(dy, unit.pop("pool_idx"), tuple(unit["y"].shape))
            _Unit.bwd(unit, dy, need_dx=False)
            return
        dz = ops.maxpool_hw_bwd(dy, rec["pool_idx"], rec["pool_in"])
        _Unit.bwd(saved.pop(), dz, need_dx=False)


class VideoModelStem(nn.Module):
    def __init__(self, cins, couts, kts, eps, mom):
        super().__init__()
        self.num_pathways = len(cins)
        for p in range(self.num_pathways):
            self.add_module(f"pathway{p}_stem", ResNetBasicStem(cins[p], couts[p], kts[p], eps, mom))


class FuseFastToSlow(nn.Module):
    def __init__(self, cfast, ratio, ksz, alpha, eps, mom):
        super().__init__()
        self.conv_f2s = Conv3dP(cfast, cfast * ratio, (ksz, 1, 1), (alpha, 1, 1), (ksz // 2, 0, 0))
        self.bn = BN3dP(cfast * ratio, eps, mom)


class BottleneckTransform(nn.Module):
    def __init__(self, cin, cout, cinner, tk, stride, eps, mom, zero_final):
        super().__init__()
        self.a = Conv3dP(cin, cinner, (tk, 1, 1), (1, 1, 1), (tk // 2, 0, 0))
        self.a_bn = BN3dP(cinner, eps, mom)
        self.b = Conv3dP(cinner, cinner, (1, 3, 3), (1, stride, stride), (0, 1, 1))
        self.b_bn = BN3dP(cinner, eps, mom)
        self.c = Conv3dP(cinner, cout, (1, 1, 1))
        self.c_bn = BN3dP(cout, eps, mom, zero_init=zero_final)


class ResBlock(nn.Module):
    def __init__(self, cin, cout, cinner, tk, stride, eps, mom, zero_final):
        super().__init__()
        self.has_sc = cin != cout or stride != 1
        if self.has_sc:
            self.branch1 = Conv3dP(cin, cout, (1, 1, 1), (1, stride, stride))
            self.branch1_bn = BN3dP(cout, eps, mom)
        self.branch2 = BottleneckTransform(cin, cout, cinner, tk, stride, eps, mom, zero_final)

    # Eval only: the identity chain of a stage kept in fp32 (ops.residual_add_f32; north_star's "logits within 1e-3").
    # The block's bf16 output carries the fp32 stream as the attribute `_vs_f32` for the next block of the stage.
    # Costs one element-wise pass per block (the c unit's residual + ReLU epilogue becomes that pass).  VS_RESIDUAL_FP32=1.
    residual_fp32 = os.environ.get("VS_RESIDUAL_FP32", "0") == "1"
    # Eval only: conv b and conv c of the fast pathway's 8 / 16 / 32-channel bottlenecks (res2 - res4) as one launch
    # (ops.conv_fwd_bc: the inner tensor stays in LDS).  VS_EVAL_FUSE_BC=0: two launches (A/B switch).
    fuse_bc = os.environ.get("VS_EVAL_FUSE_BC", "1") != "0"

    # Train: the b -> c edge without the stored activation (apply on load, `_Unit.fwd(no_apply / x_affine)`) where the
    # c unit's forward and weight-gradient plans have the fragment transform (slow pathway).  VS_TRAIN_AOL=0 / 1.
    aol = os.environ.get("VS_TRAIN_AOL", "0") == "1"

    def _aol_ok(self, a, saved):
        b2 = self.branch2
        if (not ResBlock.aol or saved is None or not a.is_cuda or _Unit.trace is not None or _Unit.wgrad_batch is not None
                or _Unit.split_wgrad_reduce or b2.b.bias is not None or b2.c.bias is not None
                or b2.c.cin_pad != b2.c.cin):
            return False
        key = ("aol",) + tuple(a.shape)
        hit = self.__dict__.setdefault("_bc_ok", {}).get(key)
        if hit is None:
            yb_shape = ops.conv_out_shape(a.shape, b2.b.cout, b2.b.k, b2.b.s, b2.b.p)
            probe = ops.new_act(*yb_shape, device=a.device)  # (shape / pitch carrier for the plan query)
            hit = self._bc_ok[key] = ops.conv_aol_ok(probe, b2.c.cout)
        return hit

    def _bc_fusable(self, a):
        b2 = self.branch2
        if (not ResBlock.fuse_bc or _Unit.split_weights or ResBlock.residual_fp32 or not a.is_cuda
                or b2.b.bias is not None or b2.c.bias is not None):
            return False
        key = tuple(a.shape)
        hit = self.__dict__.setdefault("_bc_ok", {}).get(key)
        if hit is None:
            hit = self._bc_ok[key] = ops.conv_fwd_bc_fusable(a, b2.b.w_bf16, b2.b.k, b2.b.s, b2.b.p, b2.c.cout)
        return hit

    def fwd(self, x, out, train, saved):
        b2 = self.branch2
        sc = x
        sc_aff = None
        if self.has_sc:
            if (train and ResBlock.fuse_sc_apply and _Unit.trace is None and x.is_cuda
                    and ops.bn_apply2_ok(self.branch1.cout)):
                # the shortcut's BN is applied inside the block's last apply pass: its normalised output -- a block-
                # output-sized tensor -- is neither written nor re-read (same bits as the two passes)
                sc_aff = _Unit.fwd(self.branch1, self.branch1_bn, x, False, train=True, saved=saved, defer_apply=True)
            else:
                sc = _Unit.fwd(self.branch1, self.branch1_bn, x, False, train=train, saved=saved)
        a = _Unit.fwd(b2.a, b2.a_bn, x, True, train=train, saved=saved)
        if sc_aff is not None:
            b = _Unit.fwd(b2.b, b2.b_bn, a, True, train=True, saved=saved)
            z = _Unit.fwd(b2.c, b2.c_bn, b, True, out=out, train=True, saved=saved, residual_affine=sc_aff)
            if saved is not None and len(saved) >= 4:
                saved[-1]["sc_rec"] = saved[-4]
            return z
        if train and self._aol_ok(a, saved):
            yb, aff = _Unit.fwd(b2.b, b2.b_bn, a, True, train=True, saved=saved, no_apply=True)
            z = _Unit.fwd(b2.c, b2.c_bn, yb, True, residual=sc, out=out, train=True, saved=saved, x_affine=aff)
            if self.has_sc and saved is not None and len(saved) >= 4:
                saved[-1]["sc_rec"] = saved[-4]
            return z
        if not train and self._bc_fusable(a):
            (sb, hb), (s_c, h_c) = b2.b_bn.fold, b2.c_bn.fold
            return ops.conv_fwd_bc(a, b2.b.w_bf16, b2.b.k, b2.b.s, b2.b.p, sb, hb, b2.c.w_bf16, s_c, h_c,
                                   residual=sc, relu=True, out=out)
        b = _Unit.fwd(b2.b, b2.b_bn, a, True, train=train, saved=saved)
        if ResBlock.residual_fp32 and not train:
            branch = _Unit.fwd(b2.c, b2.c_bn, b, False, train=False)
            res = sc if self.has_sc else getattr(x, "_vs_f32", x)
            z, z32 = ops.residual_add_f32(branch, res, relu=True, out16=out)
            z._vs_f32 = z32
            return z
        z = _Unit.fwd(b2.c, b2.c_bn, b, True, residual=sc, out=out, train=train, saved=saved)
        if self.has_sc and train and saved is not None and len(saved) >= 4:
            saved[-1]["sc_rec"] = saved[-4]  # the next block's conv-a dgrad also emits the shortcut unit's BN sums
        return z

    def bwd(self, saved, dout, chain=False, carry=None):
        """chain: this block's input is the previous block's output and nothing else reads it -- the record
        on top of `saved` after this block's own is then that block's c unit (`_Unit.bwd` checks the tensor
        identity), whose BN-backward sums come out of this block's conv-a dgrad.
        carry: the stage's pending grouped weight gradients (`VideoTrunk._backward_stage`): a groupable block adds its
        items and the launch goes out every `ResBlock.group_span` blocks (two blocks' weight gradients share one grid
        better than each fills its own: tools/wgrad_group_time.py --span)."""
        rc, rb, ra = saved.pop(), saved.pop(), saved.pop()
        # The block's weight gradients as ONE grouped launch behind its last data gradient (ops.conv_wgrad_group) where
        # that wins: wide, few-position blocks (slow res3 - res5 at 8 clips per GPU: 60-140 us per block instead of
        # 85-170 as three or four launches + their slab reduces; tools/wgrad_group_time.py)
        sink = [] if self._wgrad_grouped(rc, rb, ra, saved[-1] if (self.has_sc and saved) else None, dout) else None
        # (normal path only: when a unit's backward raises, the half-filled sink is dropped with the exception -- a
        #  grouped launch from it could only mask the original error)
        res = self._bwd_units(saved, dout, chain, rc, rb, ra, sink)
        if sink:
            items = [(dy, x, c.k, c.s, c.p, c.weight.grad) for dy, x, c in sink]
            if carry is None:
                ops.conv_wgrad_group(items)
            else:
                carry["items"] += items
                carry["blocks"] += 1
                if carry["blocks"] >= ResBlock.group_span or len(carry["items"]) + 4 > ops.WGRAD_GROUP_MAX:
                    ResBlock.flush_wgrads(carry)
        return res

    @staticmethod
    def flush_wgrads(carry):
        if carry and carry["items"]:
            items = carry["items"]
            # (Measured and not kept: the grouped launch on the weight-gradient lane, sized for a fraction of the CUs, beside
            #  the next blocks' kernels -- 11.20-11.28 ms against 11.05-11.16 inline: the chip has no idle share to give it.)
            ops.conv_wgrad_group(items)
        if carry:
            carry["items"], carry["blocks"] = [], 0


    # blocks whose weight gradients share one grouped launch (VS_WGRAD_GROUP_SPAN)
    group_span = int(os.environ.get("VS_WGRAD_GROUP_SPAN", "3"))  # A/B in the step: 1 -> 2 -> 3 blocks +0.2 % each, 4 the same, 6 -0.3 %

    # Train: a block's weight gradients as one launch.  VS_WGRAD_GROUP=0: per-unit launches; VS_WGRAD_GROUP_MAXP: the
    # position count up to which a block is grouped (at 32 clips per GPU the separate launches fill the chip themselves).
    group_wgrads = os.environ.get("VS_WGRAD_GROUP", "1") != "0"
    group_max_positions = int(os.environ.get("VS_WGRAD_GROUP_MAXP", "1000000000"))
    group_min_channels = int(os.environ.get("VS_WGRAD_GROUP_MINC", "128"))  # narrower blocks: neutral (64, 32) or slower (16, 8) in the step

    def _wgrad_grouped(self, rc, rb, ra, rsc, dout):
        if (not ResBlock.group_wgrads or not dout.is_cuda or _Unit.wgrad_batch is not None or _Unit.split_wgrad_reduce
                or ResBlock.aol):
            return False
        recs = [r for r in (rc, rb, ra, rsc) if r is not None]
        key = ("wgg",) + tuple(tuple(r["y"].shape) for r in recs)
        hit = self.__dict__.setdefault("_bc_ok", {}).get(key)
        if hit is None:
            hit = True
            items = []
            for r in recs:
                c, y, x = r["conv"], r["y"], r["x"]
                if (c.is_stem or c.cin_pad != c.cin or c.bias is not None or r.get("x_affine") is not None
                        or c.cout < ResBlock.group_min_channels or not isinstance(x, torch.Tensor)
                        or not getattr(c.weight, "_vs_direct_grad", True)):
                    hit = False
                    break
                items.append((y, x, c.k, c.s, c.p, torch.empty(0)))
            if hit:
                hit = min(ops.act_rows(r["y"]) for r in recs) <= ResBlock.group_max_positions
            if hit:
                # (the plan query needs descriptors only: the raw conv output stands in for its gradient, same shape / pitch)
                probe = [(y, x, k, s_, p, torch.empty((y.shape[1], *k, x.shape[1]), dtype=torch.float32,
                                                      device=y.device).permute(0, 4, 1, 2, 3)) for y, x, k, s_, p, _ in items]
                hit = ops.conv_wgrad_group_ok(probe)
            self._bc_ok[key] = hit
        return hit

    def _bwd_units(self, saved, dout, chain, rc, rb, ra, sink):
        # The gradient over the identity / shortcut branch is dout under the block's ReLU mask.  With the mask
        # kept as bits (`zbits`, written by the forward apply) its two readers take (dout, bits) and the c unit's
        # backward apply does not write a masked copy (`dres`: one block-output-sized tensor per block).
        cbits = rc.get("zbits")  # None for channel counts whose mask is not kept as bits
        dy_sc = None
        if (self.has_sc and cbits is not None and ResBlock.fuse_sc_bwd and dout.is_cuda and saved
                and rc["conv"].bias is None and saved[-1]["conv"].bias is None and ops.bn_apply2_ok(rc["y"].shape[1])):
            # the c unit and the shortcut unit receive the same masked gradient: both finalizes first, then ONE backward
            # apply pass that reads dout and the bits once and writes both dy (bitwise the two passes)
            rsc = saved[-1]
            units = []
            for r in (rc, rsc):
                bn = r["bn"]
                if bn.weight.grad is None:
                    bn.weight.grad = torch.zeros_like(bn.weight)
                if bn.bias.grad is None:
                    bn.bias.grad = torch.zeros_like(bn.bias)
                ops.bn_bwd_sums(dout, r["y"], r["mean"], r["invstd"], cbits, bn.weight.grad, bn.bias.grad,
                                partial=r.pop("bwd_partial", None))
                units.append((r["y"], r["mean"], r["invstd"], bn.weight, bn.weight.grad, bn.bias.grad))
            dy_c, dy_sc = ops.bn_bwd_apply2(dout, cbits, units[0], units[1])
            db, g = _Unit.bwd(rc, dout, producer=rb, dy_ready=dy_c, wgrad_sink=sink)
        else:
            db, g = _Unit.bwd(rc, dout, want_dres=cbits is None, producer=rb, wgrad_sink=sink)
        da, _ = _Unit.bwd(rb, db, producer=ra, wgrad_sink=sink)
        if self.has_sc:
            rsc = saved.pop()
            if ResBlock.accumulate_shortcut:
                # conv a's (unit-stride) data gradient first; the shortcut's dgrad then accumulates into it in
                # place -- a strided shortcut touches only the positions its stride reaches, instead of writing
                # a block-input-sized tensor that is 3/4 zeros and re-reading it as a residual
                dxa, _ = _Unit.bwd(ra, da, wgrad_sink=sink)
                if cbits is not None:
                    dx, _ = _Unit.bwd(rsc, dout, dz_bits=cbits, dx_residual=dxa, inplace=True, dy_ready=dy_sc, wgrad_sink=sink)
                else:
                    dx, _ = _Unit.bwd(rsc, g, masked=True, dx_residual=dxa, inplace=True, wgrad_sink=sink)
            else:
                dx1, _ = (_Unit.bwd(rsc, dout, dz_bits=cbits, dy_ready=dy_sc, wgrad_sink=sink) if cbits is not None
                          else _Unit.bwd(rsc, g, masked=True, wgrad_sink=sink))
                dx, _ = _Unit.bwd(ra, da, dx_residual=dx1, wgrad_sink=sink)
        elif cbits is not None:
            dx, _ = _Unit.bwd(ra, da, dx_residual=dout, dx_residual_bits=cbits,
                              producer=saved[-1] if (chain and saved) else None, wgrad_sink=sink)
        else:
            dx, _ = _Unit.bwd(ra, da, dx_residual=g, producer=saved[-1] if (chain and saved) else None, wgrad_sink=sink)
        return dx

    # Train, backward: the c unit's and the shortcut unit's BN-backward apply as one pass (ops.bn_bwd_apply2): the block's
    # output gradient and its ReLU bits are read once.  VS_FUSE_SC_BWD=0: two passes.
    fuse_sc_bwd = os.environ.get("VS_FUSE_SC_BWD", "1") != "0"
    # Train: the shortcut unit's BN apply inside the c unit's apply pass (ops.bn_apply2).  VS_FUSE_SC_APPLY=0: two passes.
    fuse_sc_apply = os.environ.get("VS_FUSE_SC_APPLY", "1") != "0"

    # A/B switch (VS_ACC_SHORTCUT=0: the round-1 data flow of the shortcut's gradient)
    accumulate_shortcut = os.environ.get("VS_ACC_SHORTCUT", "1") != "0"


class Nonlocal(nn.Module):
    """slowfast `nonlocal_helper.Nonlocal` (i3d_r50_nl_8x8: Kinetics_c2_I3D_NLN_8x8_R50.yaml:25-28;
    SURVEY.md 8f row f4): theta / phi / g 1x1x1 convs with bias, phi and g on the [1,2,2] max-pooled input,
    softmax(theta.phi / sqrt(dim_inner)) ("softmax") or theta.phi / positions ("dot_product"), output conv,
    BatchNorm with zero-initialised gamma, residual.  The two batched products run per clip on the
    implicit-GEMM conv kernels with phi / g^T / g / phi^T as the "weight" operand (forward and dgrad-like
    products) and on the wgrad kernel (the products that sum over the query positions); scores and
    probabilities are bf16 like every activation of the trunk."""

    def __init__(self, dim, dim_inner, pool_size, instantiation, eps, mom):
        super().__init__()
        if instantiation not in ("softmax", "dot_product"):
            raise NotImplementedError(f"NONLOCAL.INSTANTIATION={instantiation}")
        self.dim, self.dim_inner, self.instantiation = dim, dim_inner, instantiation
        self.use_pool = pool_size is not None and any(v > 1 for v in pool_size)
        if self.use_pool and list(pool_size) != [1, 2, 2]:
            raise NotImplementedError(f"NONLOCAL.POOL={pool_size}: only [1, 2, 2] is built")
        self.conv_theta = Conv3dP(dim, dim_inner, (1, 1, 1), bias=True)
        self.conv_phi = Conv3dP(dim, dim_inner, (1, 1, 1), bias=True)
        self.conv_g = Conv3dP(dim, dim_inner, (1, 1, 1), bias=True)
        self.conv_out = Conv3dP(dim_inner, dim, (1, 1, 1), bias=True)
        self.bn = BN3dP(dim, eps, mom, zero_init=True)
        self._const = {}

    K1, S1, P0 = (1, 1, 1), (1, 1, 1), (0, 0, 0)

    def _vec(self, n, value, dev):
        key = (n, float(value), str(dev))
        t = self._const.get(key)
        if t is None:
            t = self._const[key] = torch.full((n,), float(value), dtype=torch.float32, device=dev)
        return t

    @staticmethod
    def _as_weight(a):
        """One clip's activation [1, C, T, H, W] (memory [positions][C]) as a 1x1x1 conv weight
        [Cout = positions, Cin = C] -- the same bytes."""
        _, c, t, h, w = a.shape
        return a[0].permute(1, 2, 3, 0).reshape(t * h * w, 1, 1, 1, c).permute(0, 4, 1, 2, 3)

    def _biased(self, conv, x):
        return ops.conv_fwd(x, conv.w_bf16, self.K1, self.S1, self.P0, scale=self._vec(conv.cout, 1.0, x.device),
                            shift=conv.bias.detach())[0]

    def _keys(self, a, npad):
        """Per-clip weight views [npad, ci] of a key-side activation (phi / g): the activation's own
        memory when its position count is a multiple of 8 (the conv kernels' channel granularity),
        else zero-padded copies."""
        n, ci = a.shape[0], a.shape[1]
        npos = a.shape[2] * a.shape[3] * a.shape[4]
        if npad == npos:
            return [self._as_weight(a[i:i + 1]) for i in range(n)]
        buf = torch.zeros((n, npad, ci), dtype=a.dtype, device=a.device)
        buf[:, :npos].copy_(a.permute(0, 2, 3, 4, 1).reshape(n, npos, ci))
        return [buf[i].view(npad, 1, 1, 1, ci).permute(0, 4, 1, 2, 3) for i in range(n)]

    def _score_affine(self, npos, npad, dev):
        """Epilogue of the theta.phi product: the scale of the instantiation and, for padded key columns,
        a shift that makes their softmax probability exactly 0."""
        softmax = self.instantiation == "softmax"
        key = ("aff", npos, npad, str(dev))
        t = self._const.get(key)
        if t is None:
            sc = torch.full((npad,), self.dim_inner ** -0.5 if softmax else 1.0 / npos, dtype=torch.float32,
                            device=dev)
            sh = torch.zeros(npad, dtype=torch.float32, device=dev)
            if softmax:
                sh[npos:] = -30000.0
            t = self._const[key] = (sc, sh)
        return t

    def fwd(self, x, out, train, saved):
        n, c, t, h, w = x.shape
        dev = x.device
        ci = self.dim_inner
        theta = self._biased(self.conv_theta, x)
        xp, pidx = ops.maxpool_hw2(x) if self.use_pool else (x, None)
        phi, g = self._biased(self.conv_phi, xp), self._biased(self.conv_g, xp)
        npos = xp.shape[2] * xp.shape[3] * xp.shape[4]
        npad = (npos + 7) // 8 * 8
        softmax = self.instantiation == "softmax"
        sc, sh = self._score_affine(npos, npad, dev)
        phi_w, g_w = self._keys(phi, npad), self._keys(g, npad)
        prob = ops.new_act(n, npad, t, h, w, dev)  # [clip][query position][key position]
        for i in range(n):
            ops.conv_fwd(theta[i:i + 1], phi_w[i], self.K1, self.S1, self.P0, out=prob[i:i + 1], scale=sc, shift=sh)
        if softmax:
            ops.softmax_rows_bf16(prob, n * t * h * w, npad)
        o = ops.new_act(n, ci, t, h, w, dev)
        for i in range(n):
            gt = ops.weight_transpose(g_w[i])  # [ci][key positions]
            ops.conv_fwd(prob[i:i + 1], gt, self.K1, self.S1, self.P0, out=o[i:i + 1])
        z = _Unit.fwd(self.conv_out, self.bn, o, False, residual=x, out=out, train=train, saved=saved)
        if train:
            saved.append(dict(nl=self, x=x, xp=xp, pidx=pidx, theta=theta, phi=phi, g=g, phi_w=phi_w, g_w=g_w,
                              prob=prob, npos=npos, npad=npad))
        return z

    def _bias_grad(self, conv, dy):
        if conv.bias.grad is None:
            conv.bias.grad = torch.empty_like(conv.bias)
        ops.colsum_bf16(dy, out=conv.bias.grad)

    def _wgrad(self, conv, dy, x):
        if conv.weight.grad is None:
            conv.weight.grad = torch.empty_like(conv.weight)
        ops.conv_wgrad(dy, x, self.K1, self.S1, self.P0, out=conv.weight.grad)

    @staticmethod
    def _key_rows(dw, npos, ci):
        """fp32 [npad][ci] product of the wgrad kernel -> its first npos rows, flat."""
        return dw.permute(0, 2, 3, 4, 1).reshape(-1)[: npos * ci]

    def bwd(self, saved, gout):
        rec = saved.pop()
        x, xp, theta, phi, g, prob = (rec[k] for k in ("x", "xp", "theta", "phi", "g", "prob"))
        phi_w, g_w, npos, npad = rec["phi_w"], rec["g_w"], rec["npos"], rec["npad"]
        n, c, t, h, w = x.shape
        dev = x.device
        ci = self.dim_inner
        do, dres = _Unit.bwd(saved.pop(), gout, want_dres=True)  # conv_out + BN; dres: the identity branch
        softmax = self.instantiation == "softmax"
        dprob = ops.new_act(n, npad, t, h, w, dev)
        dg = torch.empty_like(g)
        for i in range(n):
            # dprob[m][p] = sum_c do[m][c] g[p][c];  dg[p][c] = sum_m prob[m][p] do[m][c]
            ops.conv_fwd(do[i:i + 1], g_w[i], self.K1, self.S1, self.P0, out=dprob[i:i + 1])
            dw = ops.conv_wgrad(prob[i:i + 1], do[i:i + 1], self.K1, self.S1, self.P0)
            ops.cast_bf16(self._key_rows(dw, npos, ci), dg[i:i + 1])
        if softmax:
            ops.softmax_rows_bwd_bf16(prob, dprob, n * t * h * w, npad, ci ** -0.5)  # -> d(theta.phi), in place
        dtheta, dphi = torch.empty_like(theta), torch.empty_like(phi)
        sc = None if softmax else self._vec(ci, 1.0 / npos, dev)
        zero = None if softmax else self._vec(ci, 0.0, dev)
        for i in range(n):
            # dtheta[m][c] = sum_p ds[m][p] phi[p][c];  dphi[p][c] = sum_m ds[m][p] theta[m][c]
            pt = ops.weight_transpose(phi_w[i])
            ops.conv_fwd(dprob[i:i + 1], pt, self.K1, self.S1, self.P0, out=dtheta[i:i + 1], scale=sc, shift=zero)
            dw = ops.conv_wgrad(dprob[i:i + 1], theta[i:i + 1], self.K1, self.S1, self.P0)
            if not softmax:
                dw = dw * (1.0 / npos)
            ops.cast_bf16(self._key_rows(dw, npos, ci), dphi[i:i + 1])
        for conv, dy, xin in ((self.conv_theta, dtheta, x), (self.conv_phi, dphi, xp), (self.conv_g, dg, xp)):
            self._bias_grad(conv, dy)
            self._wgrad(conv, dy, xin)
        if self.use_pool:
            dxp = ops.conv_dgrad(dphi, self.conv_phi.wt(), tuple(xp.shape), self.K1, self.S1, self.P0)
            dxp = ops.conv_dgrad(dg, self.conv_g.wt(), tuple(xp.shape), self.K1, self.S1, self.P0, residual=dxp)
            dpool = ops.maxpool_hw2_bwd(dxp, rec["pidx"], tuple(x.shape))
            dx = ops.conv_dgrad(dtheta, self.conv_theta.wt(), tuple(x.shape), self.K1, self.S1, self.P0,
                                residual=dpool)
            # + the identity branch: one fused add (bn_apply with scale 1, shift 0, residual)
            return ops.bn_apply(dx, self._vec(c, 1.0, dev), self._vec(c, 0.0, dev), dres, relu=False)
        dx = ops.conv_dgrad(dtheta, self.conv_theta.wt(), tuple(x.shape), self.K1, self.S1, self.P0, residual=dres)
        dx = ops.conv_dgrad(dphi, self.conv_phi.wt(), tuple(x.shape), self.K1, self.S1, self.P0, residual=dx)
        return ops.conv_dgrad(dg, self.conv_g.wt(), tuple(x.shape), self.K1, self.S1, self.P0, residual=dx)


class ResStage(nn.Module):
    def __init__(self, cins, couts, cinners, tks, strides, nblocks, nblk_tk, eps, mom, zero_final,
                 nl_inds=None, nl_pool=None, nl_inst="dot_product"):
        super().__init__()
        self.num_pathways = len(cins)
        self.num_blocks = list(nblocks)
        self.couts = list(couts)
        self.nl_inds = [list(v) for v in nl_inds] if nl_inds is not None else [[] for _ in cins]
        for p in range(self.num_pathways):
            for i in self.nl_inds[p]:  # slowfast ResStage._construct: a Nonlocal after block i
                self.add_module(f"pathway{p}_nonlocal{i}",
                                Nonlocal(couts[p], couts[p] // 2, nl_pool[p], nl_inst, eps, mom))
            n = nblocks[p]
            tk_list = (tks[p] * n)[: nblk_tk[p]] + [1] * (n - nblk_tk[p])
            for i in range(n):
                self.add_module(
                    f"pathway{p}_res{i}",
                    ResBlock(cins[p] if i == 0 else couts[p], couts[p], cinners[p], tk_list[i],
                             strides[p] if i == 0 else 1, eps, mom, zero_final))

    def blocks(self, p):
        out = []
        for i in range(self.num_blocks[p]):
            out.append(getattr(self, f"pathway{p}_res{i}"))
            if i in self.nl_inds[p]:
                out.append(getattr(self, f"pathway{p}_nonlocal{i}"))
        return out


class PathwayPool(nn.Module):
    """`pathway{p}_pool` = MaxPool3d(k = s = [kt,1,1]) (mdl_sf_base.py:26-28, 49-51);
    identity for slowfast / slow (kt = 1)."""

    def __init__(self, kt):
        super().__init__()
        self.kt = kt

    def forward(self, x):
        return x if self.kt == 1 else ops.maxpool_t(x, self.kt)[0]


class _TrunkFn(torch.autograd.Function):
    """One autograd node for the whole trunk (manual HIP backward)."""

    @staticmethod
    def forward(ctx, trunk, _tick, *inputs):
        feats, saved = trunk._run(inputs, train=True)
        ctx.trunk, ctx.saved = trunk, saved
        ctx.n_in = len(inputs)
        return tuple(feats)

    @staticmethod
    def backward(ctx, *dfeats):
        ctx.trunk._backward(ctx.saved, [ops_ensure_act(d) for d in dfeats])
        ctx.saved = None
        return (None, None) + (None,) * ctx.n_in


def ops_ensure_act(t):
    """Gradients normally arrive as our own channels-last bf16 tensors; anything
    else (a torch op in between) is re-laid-out here."""
    if t.dtype != ops.BF16:
        t = t.to(ops.BF16)
    n, c, tt, h, w = t.shape
    if not t.permute(0, 2, 3, 4, 1).is_contiguous():
        t = t.contiguous(memory_format=torch.channels_last_3d)
        if not t.permute(0, 2, 3, 4, 1).is_contiguous():  # size-1 dims confuse the format
            t = t.permute(0, 2, 3, 4, 1).contiguous().permute(0, 4, 1, 2, 3)
    return t


class _Fork:
    """Fork / join of the two pathways onto two HIP streams.  Between two lateral connections the
    slow and the fast pathway are independent chains; the fast one (1/8 of the channels) is made of
    small, latency-bound launches that hide behind the slow pathway's when both are in flight.
    Pathway 0 stays on the caller's stream, pathway 1 runs on a side stream; `join` makes the
    caller's stream wait for the side stream (and the side stream for the caller's at the next
    `fork`), so a captured hipGraph gets two parallel branches per stage.  Tensors that cross
    streams are kept alive (`keep`) until the join: nothing is freed and re-used while the other
    stream may still read it."""

    def __init__(self, side):
        self.side = side
        self.main = None
        self.active = False

    def fork(self):
        if self.side is None:
            return
        self.main = torch.cuda.current_stream()
        self.side.wait_stream(self.main)
        self.active = True

    def on(self, p):
        import contextlib

        if self.side is None or not self.active or p == 0:
            return contextlib.nullcontext()
        return torch.cuda.stream(self.side)

    def join(self, keep=None):
        if _WgradLanes.defer_active:
            ops.wgrad_reduce_flush()  # a slab reduce still waiting on the side stream must be IN that stream before the join
        _WgradLanes.join_all()
        if self.side is None or not self.active:
            return
        self.main.wait_stream(self.side)
        self.active = False
        if isinstance(keep, list):
            keep.clear()


class VideoTrunk(nn.Module):
    """`SlowFast_FeatModel` / `ResNet_FeatModel` (mdl_sf_base.py:20-62)."""

    # run the two pathways of a multi-pathway trunk on two streams (VS_DUAL_STREAM=0: one stream)
    dual_stream = os.environ.get("VS_DUAL_STREAM", "1") != "0"
    # Fork + full join around every stage.  A one-directional variant (fork once per pass, the slow
    # pathway waits for the fast one only at the lateral connections, so the fast pathway may run
    # ahead) was measured SLOWER: train 16.85 vs 16.47 ms, forward 3.17 vs 2.96 ms at batch 8.
    _side_streams = {}

    def _fork_ctx(self, dev):
        if not (self.dual_stream and self.multi and dev.type == "cuda"):
            return _Fork(None)
        key = dev.index if dev.index is not None else torch.cuda.current_device()
        side = VideoTrunk._side_streams.get(key)
        if side is None:
            side = VideoTrunk._side_streams[key] = torch.cuda.Stream(device=dev)
        return _Fork(side)

    def __init__(self, cfg):
        super().__init__()
        arch = cfg.MODEL.ARCH
        self.arch = arch
        self.multi = arch in cfg.MODEL.MULTI_PATHWAY_ARCH
        data = getattr(cfg, "DATA", None)  # CfgNode or a plain namespace (oracle's default_sf_cfg)
        self.data_mean = tuple(getattr(data, "MEAN", (0.45, 0.45, 0.45)))
        self.data_std = tuple(getattr(data, "STD", (0.225, 0.225, 0.225)))
        self.data_reverse = bool(getattr(data, "REVERSE_INPUT_CHANNEL", False))
        self.alpha = cfg.SLOWFAST.ALPHA if self.multi else 1
        self._slow_idx_cache = {}
        self.num_pathways = 2 if self.multi else 1
        self.enable_detection = False
        eps, mom = cfg.BN.EPSILON, cfg.BN.MOMENTUM
        zf = bool(cfg.RESNET.ZERO_INIT_FINAL_BN)
        w = cfg.RESNET.WIDTH_PER_GROUP
        inner = cfg.RESNET.NUM_GROUPS * w
        depths = STAGE_DEPTH[cfg.RESNET.DEPTH]
        tk = TEMPORAL_KERNEL_BASIS[arch]
        nbt = cfg.RESNET.NUM_BLOCK_TEMP_KERNEL
        ss = cfg.RESNET.SPATIAL_STRIDES
        self.pool1 = POOL1[arch]
        if self.multi:
            binv = cfg.SLOWFAST.BETA_INV
            ratio = cfg.SLOWFAST.FUSION_CONV_CHANNEL_RATIO
            odr = binv // ratio
            fk, alpha = cfg.SLOWFAST.FUSION_KERNEL_SZ, cfg.SLOWFAST.ALPHA
            self.s1 = VideoModelStem(cfg.DATA.INPUT_CHANNEL_NUM, [w, w // binv],
                                     [tk[0][0][0], tk[0][1][0]], eps, mom)
            self.s1_fuse = FuseFastToSlow(w // binv, ratio, fk, alpha, eps, mom)
            cin_s, cin_f = w + w // odr, w // binv
            for k in range(4):
                mult = 4 * (2 ** k)
                cout_s, cout_f = w * mult, w * mult // binv
                setattr(self, f"s{k + 2}", ResStage(
                    [cin_s, cin_f], [cout_s, cout_f],
                    [inner * (2 ** k), inner * (2 ** k) // binv], tk[k + 1], ss[k],
                    [depths[k]] * 2, nbt[k], eps, mom, zf))
                if k < 3:
                    setattr(self, f"s{k + 2}_fuse",
                            FuseFastToSlow(cout_f, ratio, fk, alpha, eps, mom))
                cin_s, cin_f = cout_s + cout_s // odr, cout_f
            self.dim_out = [w * 32, w * 32 // binv]
        else:
            self.s1 = VideoModelStem(cfg.DATA.INPUT_CHANNEL_NUM, [w], [tk[0][0][0]], eps, mom)
            cin = w
            for k in range(4):
                cout = w * 4 * (2 ** k) if depths[k] > 0 else cin
                nlc = getattr(cfg, "NONLOCAL", None)
                loc = nlc.LOCATION[k] if nlc is not None else [[]]
                setattr(self, f"s{k + 2}", ResStage(
                    [cin], [cout], [inner * (2 ** k)], tk[k + 1], ss[k], [depths[k]], nbt[k],
                    eps, mom, zf, nl_inds=loc,
                    nl_pool=(nlc.POOL[k] if (nlc is not None and any(loc)) else None),
                    nl_inst=getattr(nlc, "INSTANTIATION", "dot_product") if nlc is not None else "dot_product"))
                cin = cout
            self.dim_out = [cin]
        for p in range(self.num_pathways):
            self.add_module(f"pathway{p}_pool", PathwayPool(self.pool1[p][0]))
        self._weights_version = None
        self._folds_version = None
        self._stats_epoch = 0  # bumped whenever a train-mode pass rewrites running stats
        self.debug_taps = None  # set to a dict to record the activations after every stage
        self._wgrad_batch, self._reduce_pending = None, False
        self.num_classes = int(getattr(cfg.MODEL, "NUM_CLASSES", 400))

    def reference_only_params(self):
        """Parameters the reference's `sf_mdl` owns and this trunk never builds: the upstream
        classification head `head.projection` (Linear(sum(dim_out), NUM_CLASSES)), constructed by
        `SlowFast` / `ResNet` but never executed by `forward_features` (`mdl_sf_base.py:21-34`).
        They sit in every reference checkpoint and in the reference's optimizer parameter order
        (`checkpoint.py`, `optim.reference_param_order`)."""
        return [("head.projection.weight", (self.num_classes, sum(self.dim_out))),
                ("head.projection.bias", (self.num_classes,))]

    # ---- weights ----------------------------------------------------------------
    def _convs(self):
        return [m for m in self.modules() if isinstance(m, Conv3dP)]

    def refresh_weights(self):
        """(Re)build the bf16 kernel-layout copies of every conv weight.  Call after
        an optimizer step or a state_dict load; `forward_features` calls it
        lazily when the parameters' version counters moved."""
        for c in self._convs():
            c.refresh()
        self._weights_version = self._version_key()

    def _version_key(self):
        return tuple((c.weight._version, c.weight.data_ptr()) for c in self._convs())

    def _bns(self):
        return [m for m in self.modules() if isinstance(m, BN3dP)]

    def _fold_key(self):
        return (self._stats_epoch,) + tuple(
            (b.weight._version, b.bias._version, b.running_mean._version, b.running_var._version,
             b.weight.data_ptr()) for b in self._bns())

    def _ensure_folds(self):
        """Eval mode: BN folds to a per-channel (scale, shift) applied in the conv epilogue;
        recomputed only when parameters / running statistics changed."""
        key = self._fold_key()
        if self._folds_version != key:
            for b in self._bns():
                sc, sh, _, _ = ops.bn_finalize(None, 0, b.weight, b.bias, b.running_mean,
                                               b.running_var, b.momentum, b.eps, train=False)
                b.fold_raw = (sc, sh)
                b.fold = (sc, sh) if b.wround_bias is None else (sc, sh - sc * b.wround_bias)
            self._folds_version = key

    def calibrate_weight_rounding(self, x):
        """Eval-mode bias correction for the bf16 rounding of the convolution weights (north_star: "logits within 1e-3
        of reference" on the arithmetic that is timed).  x: calibration clips, a list like forward_features' input --
        NOT the clips that are evaluated, but clips OF THE SAME DISTRIBUTION (a calibration on other data is a bias, not a
        correction: profiles/parity_eval.json "robustness").  One eval forward pass in which every convolution (since
        round 6 the two Cin = 3 stems too: a video's brightness is a channel mean) measures its input's channel means and stores
        the per-channel constant its weight rounding adds (`BN3dP.wround_bias`); from then on the folded BN shift of
        every eval forward carries the correction -- no launch, byte or FLOP more per forward.  Measured on one
        224^2 SlowFast-R50 clip (tools/bias_correction_probe.py, CPU oracle: the weight-rounding error of the logits
        3.9e-3 -> 5.7e-4 with 2 calibration clips, 6 clips the same; tests/test_gpu_parity_full.py for the HIP path).
        The correction belongs to the weights it was measured with: an optimizer step or a state_dict load drops it
        (`reset_weight_rounding`); training never uses it (batch statistics remove a per-channel constant exactly)."""
        if self.training:
            raise ops._lib.VsError("calibrate_weight_rounding is an eval-mode pass (call .eval() first)")
        self.reset_weight_rounding()
        fb, sw = ResBlock.fuse_bc, _Unit.split_weights
        ResBlock.fuse_bc, _Unit.split_weights = False, False  # every convolution as a launch of its own: its input exists
        _Unit.calib = {}
        try:
            with torch.no_grad():
                self.forward_features(x)
        finally:
            n = len(_Unit.calib)
            _Unit.calib = None
            ResBlock.fuse_bc, _Unit.split_weights = fb, sw
        self._wround_key = self._version_key()
        self._folds_version = None  # re-fold (the corrected folds of the pass are what a re-fold produces)
        return n

    def reset_weight_rounding(self):
        for b in self._bns():
            b.wround_bias = None
        self._wround_key = None
        self._folds_version = None

    # ---- forward ------------------------------------------------------------------
    def forward_features(self, x):
        """x: list of NCDHW tensors ([slow, fast] or [fast]) -> list of feature maps
        (bf16, logical NCDHW, channels-last memory)."""
        if not x[0].is_cuda:
            raise ops._lib.VsError("VideoTrunk runs on the HIP kernels only (GPU tensors required)")
        if self._weights_version != self._version_key():
            self.refresh_weights()
            if getattr(self, "_wround_key", None) is not None and self._wround_key != self._weights_version:
                self.reset_weight_rounding()  # measured with other weights
        if self.training and torch.is_grad_enabled():
            tick = torch.zeros(1, device=x[0].device, requires_grad=True)
            return list(_TrunkFn.apply(self, tick, *x))
        feats, _ = self._run(x, train=self.training)
        return feats

    def forward(self, x, bboxes=None):
        return self.forward_features(x)

    def _slow_index(self, t, dev):
        """`pack_pathway_output` (utils/video_utils.py:59-65): linspace(0, T-1, T // alpha).long()."""
        key = (t, str(dev))
        if key not in self._slow_idx_cache:
            idx = torch.linspace(0, t - 1, t // self.alpha).long().to(torch.int32)
            self._slow_idx_cache[key] = idx.to(dev)
        return self._slow_idx_cache[key]

    def _run(self, inputs, train):
        saved = [] if train else None
        P = self.num_pathways
        xin, shapes = [], []
        dev = inputs[0].device
        if train:
            bns = [m.num_batches_tracked for m in self.modules() if isinstance(m, BN3dP)]
            torch._foreach_add_(bns, 1)
            self._stats_epoch += 1
        else:
            self._ensure_folds()
        # Fork BEFORE packing: each pathway's packed input must belong to the stream that reads it
        # (a buffer from the main stream's pool, freed when the fast stem's backward record is
        # dropped, was re-used by the slow pathway while the fast stem's wgrad was still reading
        # it: caught by test_hipgraph_two_stream_step_is_bitwise_the_one_stream_eager_step)
        par = self._fork_ctx(dev)
        s1_buf = None
        if self.multi:  # (the s1 concat buffer: before the fork, for the reason given at the stage loop below)
            st0 = self.s1.pathway0_stem
            sh0 = (tuple(inputs[0].shape) if inputs[0].dtype != torch.uint8 else
                   (inputs[0].shape[0], 3, int(self._slow_index(inputs[0].shape[1], dev).numel()), inputs[0].shape[2],
                    inputs[0].shape[3]))
            ho0, wo0 = (sh0[3] + 6 - 7) // 2 + 1, (sh0[4] + 6 - 7) // 2 + 1
            s1_buf = ops.new_act(sh0[0], st0.conv.cout + self.s1_fuse.conv_f2s.cout, sh0[2],
                                 (ho0 + 2 - 3) // 2 + 1, (wo0 + 2 - 3) // 2 + 1, dev)
        par.fork()
        if inputs[0].dtype == torch.uint8:
            # uint8 frames [N, T, H, W, 3] (what the loader's PIL step produces): normalise, pack and
            # gather the slow pathway's frames on the GPU (vs_frames_u8_pack), one launch per pathway
            fr = inputs[0]
            n, t, h, w, _ = fr.shape
            for p in range(P):
                tidx = None
                if self.multi and p == 0:
                    tidx = self._slow_index(t, dev)
                stem = getattr(self.s1, f"pathway{p}_stem").conv.is_stem
                with par.on(p):
                    y = ops.frames_u8_pack(fr, 4 if stem else 8, tidx, self.data_mean, self.data_std,
                                           self.data_reverse)
                xin.append((y, None) if stem else y)
                shapes.append((n, 3, t if tidx is None else int(tidx.numel()), h, w))
        else:
            for p, t in enumerate(inputs):
                with par.on(p):
                    if getattr(self.s1, f"pathway{p}_stem").conv.is_stem:
                        xin.append((ops.pack_input(t, 4), None))
                    else:
                        xin.append(ops.pack_input(t))
                shapes.append(tuple(t.shape))
        # ---- s1 (+ fuse): stems write straight into the concat buffer of the slow path
        cur = []
        for p in range(P):
            stem = getattr(self.s1, f"pathway{p}_stem")
            n, _, t, h, w = shapes[p]
            ho, wo = (h + 6 - 7) // 2 + 1, (w + 6 - 7) // 2 + 1
            hp, wp = (ho + 2 - 3) // 2 + 1, (wo + 2 - 3) // 2 + 1
            c = stem.conv.cout
            with par.on(p):  # allocate on the stream that writes the buffer first
                if self.multi and p == 0:
                    buf = s1_buf
                    assert tuple(buf.shape) == (n, c + self.s1_fuse.conv_f2s.cout, t, hp, wp)
                    out = ops.channel_slice(buf, 0, c)
                else:
                    buf = out = ops.new_act(n, c, t, hp, wp, dev)
                stem.fwd(xin[p], out, train, saved)
            cur.append(buf)
        if self.multi and VideoTrunk.fuse_on_fast:
            with par.on(1):  # the lateral connection on the fast pathway's stream: off the slow pathway's chain
                self._fuse_fwd(self.s1_fuse, cur, train, saved)
        par.join()
        if self.multi and not VideoTrunk.fuse_on_fast:
            self._fuse_fwd(self.s1_fuse, cur, train, saved)
        if self.debug_taps is not None:
            self.debug_taps["s1"] = [c.float().cpu() for c in cur]
        for k in range(2, 6):
            stage = getattr(self, f"s{k}")
            fuse = getattr(self, f"s{k}_fuse", None) if self.multi else None
            nxt = []
            buf = None
            if fuse is not None:
                # The slow pathway's concat buffer [stage output | lateral connection] is allocated BEFORE the fork: the
                # lateral convolution writes its slice from the fast pathway's stream, which is ordered behind the
                # caller's stream only up to the fork.  Allocated inside the stage (round 5 and before), the buffer could
                # land in memory an intermediate tensor of THIS stage had just been freed from on the host while its
                # readers were still queued on the slow pathway's stream -- and the lateral write overtook them whenever
                # the GPU ran behind the host (found in round 6: the first eval forward after a two-clip calibration pass
                # differed from every later one; bit-identical with a synchronisation in between).
                b0 = stage.blocks(0)[0].branch2.b
                ys = ops.conv_out_shape(cur[0].shape, stage.couts[0], (1, 1, 1), (1, b0.s[1], b0.s[2]), (0, 0, 0))
                buf = ops.new_act(ys[0], ys[1] + fuse.conv_f2s.cout, ys[2], ys[3], ys[4], dev)
            par.fork()
            for p in range(P):
              with par.on(p):
                x = cur[p]
                blocks = stage.blocks(p)
                for i, blk in enumerate(blocks):
                    last = i == len(blocks) - 1
                    if last and fuse is not None and p == 0:
                        blk.fwd(x, ops.channel_slice(buf, 0, ys[1]), train, saved)
                        x = buf
                    else:
                        x = blk.fwd(x, None, train, saved)
                nxt.append(x)
            if fuse is not None and VideoTrunk.fuse_on_fast:
                with par.on(1):
                    self._fuse_fwd(fuse, nxt, train, saved)
            par.join(keep=cur)
            cur = nxt
            if fuse is not None and not VideoTrunk.fuse_on_fast:
                self._fuse_fwd(fuse, cur, train, saved)
            if k == 2:
                for p in range(P):
                    kt = self.pool1[p][0]
                    if kt > 1:
                        with par.on(p):
                            y, idx = ops.maxpool_t(cur[p], kt, want_idx=train)
                        if train:
                            saved.append(dict(tpool_idx=idx, tpool_in=tuple(cur[p].shape), kt=kt))
                        cur[p] = y
            if self.debug_taps is not None:
                self.debug_taps[f"s{k}"] = [c.float().cpu() for c in cur]
        return cur, saved

    def _fuse_fwd(self, fuse, cur, train, saved):
        slow_buf, fast = cur
        cf = fuse.conv_f2s.cout
        cs = slow_buf.shape[1] - cf
        _Unit.fwd(fuse.conv_f2s, fuse.bn, fast, True, out=ops.channel_slice(slow_buf, cs, cf),
                  train=train, saved=saved)

    # ---- backward (mirror of _run, popping `saved`) ----------------------------------
    # Backward segments, in execution order.  With `defer_backward` the autograd node only parks
    # the incoming gradients and the caller runs the segments itself (bench.py captures each one
    # in its own hipGraph and all-reduces a finished bucket of gradients behind the next segment).
    BWD_SEGMENTS = ("s5", "s4", "rest")
    defer_backward = False
    _deferred = None

    def _backward(self, saved, dfeats):
        st = {"saved": saved, "d": list(dfeats)}
        if self.defer_backward:
            self._deferred = st
            return
        for seg in self.BWD_SEGMENTS:
            self._backward_segment(st, seg)

    def run_backward_segment(self, seg):
        if self._deferred is None:
            raise ops._lib.VsError("run_backward_segment without a deferred backward")
        self._backward_segment(self._deferred, seg)
        if seg == self.BWD_SEGMENTS[-1]:
            self._deferred = None

    def backward_segment_modules(self, seg):
        """Modules whose parameter gradients are complete once `seg` has run."""
        names = {"s5": ["s5"], "s4": ["s4", "s4_fuse"],
                 "rest": ["s1", "s1_fuse", "s2", "s2_fuse", "s3", "s3_fuse"]}[seg]
        return [getattr(self, n) for n in names if hasattr(self, n)]

    # One slab-reduce launch per backward segment instead of one behind every weight gradient (108 launches of
    # ~6 us per SlowFast-R50 step).  Measured SLOWER (A/B on one box, 30 steps: 14.00 ms batched vs 13.77 ms): the
    # per-layer reduce reads slabs its wgrad wrote microseconds earlier (L2 / Infinity Cache), the batched one
    # reads 1.45 GB back from HBM and competes with the BN passes.  Kept behind VS_WGRAD_BATCH=1.
    batch_wgrads = os.environ.get("VS_WGRAD_BATCH", "0") == "1"
    _reduce_streams = {}

    def _flush_wgrads(self, last):
        """Sum the pending weight-gradient slabs.  The launch goes to a side stream beside the next segment; it
        is joined at the end of the backward pass -- or at once when the segments are driven from outside
        (`defer_backward`: the caller all-reduces the segment's gradients right behind it)."""
        batch = self._wgrad_batch
        if batch is None or (not batch.pending and not self._reduce_pending):
            return
        if ops._WHATIF & 4:  # timing experiment only: the slabs are never summed (garbage gradients)
            batch.pending.clear() if hasattr(batch.pending, "clear") else None
            return
        main = torch.cuda.current_stream()
        key = main.device.index
        rs = VideoTrunk._reduce_streams.get(key)
        if rs is None:
            rs = VideoTrunk._reduce_streams[key] = torch.cuda.Stream(device=main.device)
        if batch.pending:
            rs.wait_stream(main)
            with torch.cuda.stream(rs):
                batch.flush()
            self._reduce_pending = True
        if self._reduce_pending and (last or self.defer_backward):
            main.wait_stream(rs)
            self._reduce_pending = False

    def _backward_segment(self, st, seg):
        if self.batch_wgrads and st["d"][0].is_cuda:
            if self._wgrad_batch is None:
                self._wgrad_batch = ops.WgradBatch()
            _Unit.wgrad_batch = self._wgrad_batch
        # Slab reduces of weight gradients issued on this thread's streams wait for the next unit's BN-backward finalize
        # and share its launch (ops.REDUCE_MERGE, vs_wgrad_reduce_defer): ~70 launches fewer on the step's chain.
        # Whatever is still pending at the end of the segment is launched there (the caller may all-reduce next).
        merge = ops.REDUCE_MERGE and st["d"][0].is_cuda and not self.batch_wgrads
        if merge:
            ops.wgrad_reduce_defer(1)
            _WgradLanes.defer_active = True
        try:
            self._backward_segment_body(st, seg)
        finally:
            _Unit.wgrad_batch = None
            if merge:
                _WgradLanes.defer_active = False
                ops.wgrad_reduce_defer(0)  # flushes
        self._flush_wgrads(last=seg == self.BWD_SEGMENTS[-1])

    def _backward_segment_body(self, st, seg):
        for k in {"s5": (5,), "s4": (4,), "rest": (3, 2)}[seg]:
            self._backward_stage(st, k)
        if seg == "rest":
            saved, d = st["saved"], st["d"]
            late = None
            if self.multi:
                d, late = self._fuse_bwd(saved, d, defer=VideoTrunk.fuse_on_fast)
            par = self._fork_ctx(d[0].device)
            d_in = list(d)
            par.fork()
            for p in reversed(range(self.num_pathways)):
                with par.on(p):
                    if p == 1 and late is not None:
                        d[1] = late()
                    getattr(self.s1, f"pathway{p}_stem").bwd(saved, d[p])
            par.join(keep=d_in)
            assert not saved, "trunk backward did not consume every saved record"

    def _backward_stage(self, st, k):
        saved, d = st["saved"], st["d"]
        P = self.num_pathways
        stage = getattr(self, f"s{k}")
        fuse = getattr(self, f"s{k}_fuse", None) if self.multi else None
        if k == 2:
            for p in reversed(range(P)):
                if self.pool1[p][0] > 1:
                    rec = saved.pop()
                    d[p] = ops.maxpool_t_bwd(d[p], rec["tpool_idx"], rec["tpool_in"], rec["kt"])
        late = None
        if fuse is not None:
            d, late = self._fuse_bwd(saved, d, defer=VideoTrunk.fuse_on_fast)
        par = self._fork_ctx(d[0].device)
        d_in = list(d)  # keep the incoming gradients alive until both streams are done with them
        par.fork()
        for p in reversed(range(P)):
            with par.on(p):
                if p == 1 and late is not None:
                    d[1] = late()  # the lateral connection's backward, on the fast pathway's stream
                g = d[p]
                blks = stage.blocks(p)
                carry = {"items": [], "blocks": 0}  # grouped weight gradients waiting for their launch (this stream's)
                for i in reversed(range(len(blks))):
                    # chain: the block's input is the previous block's output and has no other consumer
                    # (a stage's first block shares its input with the lateral connection / the shortcut)
                    if isinstance(blks[i], ResBlock):
                        g = blks[i].bwd(saved, g, chain=i > 0, carry=carry)
                    else:
                        ResBlock.flush_wgrads(carry)
                        g = blks[i].bwd(saved, g)
                ResBlock.flush_wgrads(carry)
                d[p] = g
        par.join(keep=d_in)
        st["d"] = d

    # The lateral connections (FuseFastToSlow: conv_f2s + BN + ReLU of the FAST pathway's output, written into the slow
    # pathway's concat buffer) run on the fast pathway's stream -- forward before the stage's join, backward behind the
    # next fork -- instead of on the caller's stream between two stages: nothing of the slow pathway's chain waits for
    # them but the join it waits at anyway, and their strided data gradients (55 / 31 / 20 / 20 us) leave the chain.
    # Same launches, same bits.  VS_FUSE_ON_FAST=0: on the caller's stream.
    fuse_on_fast = os.environ.get("VS_FUSE_ON_FAST", "1") != "0"

    def _fuse_bwd(self, saved, d, defer=False):
        """-> ([slow gradient, fast gradient], None), or with `defer` ([slow gradient, None], thunk): the thunk runs the
        lateral unit's backward (on whatever stream is current when it is called) and returns the fast gradient."""
        d_cat, d_fast = d
        rec = saved.pop()
        cf = rec["conv"].cout
        cs = d_cat.shape[1] - cf
        run = lambda: _Unit.bwd(rec, ops.channel_slice(d_cat, cs, cf), dx_residual=d_fast)[0]
        if defer:
            return [ops.channel_slice(d_cat, 0, cs), None], run
        return [ops.channel_slice(d_cat, 0, cs), run()], None
