"""Tensor-level wrappers over the C-ABI (include/vidsitu_hip.h).

PyTorch is used for device memory and streams only; every function here ends in
a HIP kernel of libvidsitu_hip.so and raises if the library or a GPU tensor is
missing -- there is no CPU / eager fallback on the product path.

Activations: bf16 tensors of logical shape [N, C, T, H, W] whose memory is
channels-last (NDHWC), optionally a channel slice of a wider buffer (row pitch
`ld` = the buffer's channel count), which is how the slow/fast lateral concat
(`FuseFastToSlow`) is done without a copy.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import (  # noqa: F401  (re-exported flags)
    VS_CONV_AFFINE,
    VS_CONV_NAIVE,
    VS_CONV_RELU,
    VS_CONV_RESIDUAL,
    VS_CONV_STATS,
    ConvDesc,
)

BF16 = torch.bfloat16


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.VsError("HIP op called with a non-GPU tensor (no CPU fallback exists)")
    return C.c_void_p(t.data_ptr())


# ----------------------------------------------------------------------------
# activation helpers
# ----------------------------------------------------------------------------
def new_act(n, c, t, h, w, device, ctot=None, c_off=0, dtype=BF16, zero=False):
    """[n,c,t,h,w] channels-last activation; with ctot, a slice of a wider buffer."""
    ctot = c if ctot is None else ctot
    alloc = torch.zeros if zero else torch.empty
    buf = alloc((n, t, h, w, ctot), dtype=dtype, device=device)
    return buf.permute(0, 4, 1, 2, 3)[:, c_off : c_off + c]


def channel_slice(x, c_off, c):
    return x[:, c_off : c_off + c]


def act_ld(x):
    """Row pitch (elements) of a channels-last activation; validates the layout."""
    n, c, t, h, w = x.shape
    s = x.stride()
    ld = s[4] if w > 1 else (s[3] if h > 1 else (s[2] if t > 1 else s[0]))
    ok = (c == 1 or s[1] == 1) and ld >= c
    ok = ok and (w == 1 or s[4] == ld) and (h == 1 or s[3] == w * ld)
    ok = ok and (t == 1 or s[2] == h * w * ld) and (n == 1 or s[0] == t * h * w * ld)
    if not ok or x.dtype != BF16:
        raise _lib.VsError(f"not a bf16 channels-last activation: shape {tuple(x.shape)} stride {s}")
    if ld % 8 or (x.storage_offset() % 8):
        raise _lib.VsError("activation pitch / channel offset must be multiples of 8")
    return ld


def act_rows(x):
    n, c, t, h, w = x.shape
    return n * t * h * w


def pack_input(x, cpad=8):
    """NCDHW f32/bf16 (any strides are made dense first) -> NDHWC bf16, C padded (4: stem
    kernel layout, 8: generic implicit-GEMM layout)."""
    x = x.contiguous()
    n, c, t, h, w = x.shape
    y = new_act(n, cpad, t, h, w, x.device)
    _lib.call(
        "vs_pack_input", _ptr(x), int(x.dtype == BF16), _ptr(y), n, c, t, h, w, cpad, _stream()
    )
    return y


def frames_u8_pack(frames, cpad, t_index=None, mean=(0.45, 0.45, 0.45), std=(0.225, 0.225, 0.225),
                   reverse=False):
    """uint8 [N, T, H, W, 3] frames -> normalised bf16 activation [N, cpad, Tout, H, W]
    (channels-last memory); t_index: int32 device tensor of frame indices (slow pathway) or None."""
    if frames.dtype != torch.uint8 or frames.dim() != 5 or frames.shape[-1] != 3:
        raise _lib.VsError("frames_u8_pack expects uint8 [N, T, H, W, 3]")
    frames = frames.contiguous()
    n, t, h, w, _ = frames.shape
    tout = t if t_index is None else int(t_index.numel())
    y = new_act(n, cpad, tout, h, w, frames.device)
    m3 = (C.c_float * 3)(*[float(a) for a in mean])
    s3 = (C.c_float * 3)(*[float(a) for a in std])
    _lib.call("vs_frames_u8_pack", _ptr(frames), _ptr(t_index), _ptr(y), n, t, tout, h, w, cpad,
              C.cast(m3, C.c_void_p), C.cast(s3, C.c_void_p), int(reverse), _stream())
    return y


# ----------------------------------------------------------------------------
# convolution
# ----------------------------------------------------------------------------
def conv_out_shape(xs, cout, k, s, p):
    n, _, t, h, w = xs
    return (
        n,
        cout,
        (t + 2 * p[0] - k[0]) // s[0] + 1,
        (h + 2 * p[1] - k[1]) // s[1] + 1,
        (w + 2 * p[2] - k[2]) // s[2] + 1,
    )


def make_desc(xs, x_ld, ys, y_ld, k, s, p, flags=0, res_ld=0):
    d = ConvDesc()
    d.N, d.Cin, d.Ti, d.Hi, d.Wi = xs[0], xs[1], xs[2], xs[3], xs[4]
    d.Cout, d.To, d.Ho, d.Wo = ys[1], ys[2], ys[3], ys[4]
    d.kT, d.kH, d.kW = k
    d.sT, d.sH, d.sW = s
    d.pT, d.pH, d.pW = p
    d.x_ld, d.y_ld, d.res_ld, d.flags = x_ld, y_ld, res_ld, flags
    return d


def check_weight(w, cout, cin, k):
    """w: bf16 [Cout,Cin,kT,kH,kW] whose memory is [Cout][kT][kH][kW][Cin]."""
    if tuple(w.shape) != (cout, cin, *k) or w.dtype != BF16:
        raise _lib.VsError(f"weight shape/dtype mismatch {tuple(w.shape)} {w.dtype}")
    if not w.permute(0, 2, 3, 4, 1).is_contiguous():
        raise _lib.VsError("conv weight must be channels-last ([Cout][taps][Cin] in memory)")


TILE_CFGS = [(128, 128), (64, 128), (128, 64), (64, 64), (256, 32), (256, 16), (256, 128), (128, 256)]
_tune = None


def tile_flag(kind, M, ncols, K, k, s, force=None):
    """flags bits 8..11: a forced tile id, or 0 = the library's own plan (pick_tile)."""
    return (force + 1) << 8 if force is not None else 0


def conv_fwd(x, w, k, s, p, out=None, scale=None, shift=None, residual=None, relu=False,
             stats=False, naive=False, tile=None, dbg=0, splitk=False, ring=0, halo=True, pw=True, deep=True,
             splitk_il=None):
    """y = conv3d(x, w) [*scale+shift] [+residual] [relu]; optional BN-stat partials.
    Returns (y, partials|None)."""
    cout = w.shape[0]
    ys = conv_out_shape(x.shape, cout, k, s, p)
    if out is None:
        out = new_act(*ys, device=x.device)
    elif tuple(out.shape) != ys:
        raise _lib.VsError(f"conv out shape {tuple(out.shape)} != {ys}")
    check_weight(w, cout, x.shape[1], k)
    flags = 0
    if scale is not None:
        flags |= VS_CONV_AFFINE
    if residual is not None:
        flags |= VS_CONV_RESIDUAL
        if tuple(residual.shape) != ys:
            raise _lib.VsError("residual shape mismatch")
    if relu:
        flags |= VS_CONV_RELU
    if stats:
        flags |= VS_CONV_STATS
    if naive:
        flags |= VS_CONV_NAIVE
    flags |= tile_flag("f", ys[0] * ys[2] * ys[3] * ys[4], cout, x.shape[1] * k[0] * k[1] * k[2], k, s,
                       tile)
    flags |= (dbg & 7) << 12  # diagnostic ablation builds (wrong results), tools/ only
    if splitk:
        flags |= 1 << 15  # VS_CONV_SPLITK
    flags |= (ring & 7) << 16  # VS_CONV_RING
    if not halo:
        flags |= 1 << 21  # VS_CONV_NOHALO
    elif halo == "force":
        flags |= 1 << 22  # VS_CONV_FORCEHALO
    if not pw:
        flags |= 1 << 23  # VS_CONV_NOPW
    elif pw == "force":
        flags |= 1 << 24  # VS_CONV_FORCEPW
    if not deep:
        flags |= 1 << 27  # VS_CONV_NODEEP
    elif deep == "force":
        flags |= 1 << 28  # VS_CONV_FORCEDEEP
    if splitk_il is not None:  # in-launch split-K: "force" / True = wherever eligible, False = never, None = the plan
        flags |= (1 << 29) if splitk_il else (1 << 30)  # VS_CONV_SPLITK_IL / VS_CONV_NOSPLITK_IL
    d = make_desc(x.shape, act_ld(x), ys, act_ld(out), k, s, p, flags,
                  act_ld(residual) if residual is not None else 0)
    partials = None
    if stats:
        rows = _lib.load().vs_conv_stats_rows(C.byref(d))
        partials = torch.empty((rows, 2, cout), dtype=torch.float32, device=x.device)
    need = _lib.load().vs_conv_workspace_bytes(C.byref(d), 0)
    ws = _workspace(need, x.device, "splitk") if need else None
    _lib.call("vs_conv_fwd", _ptr(x), _ptr(w), _ptr(out), C.byref(d), _ptr(scale), _ptr(shift),
              _ptr(residual), _ptr(partials), _ptr(ws),
              C.c_size_t(ws.numel() if ws is not None else 0), _stream())
    return out, partials


_K1, _S1, _P0 = (1, 1, 1), (1, 1, 1), (0, 0, 0)
VS_CONV_NODEEP = 1 << 27  # include/vidsitu_hip.h


def conv_aol_ok(x, cout, stats=True):
    """True if a 1x1x1 convolution of x to cout channels can take x as its producer's RAW output (conv_fwd_aol) AND
    its weight gradient can (conv_wgrad_aol): both or neither, the activation is then never stored."""
    ys = conv_out_shape(x.shape, cout, _K1, _S1, _P0)
    d = make_desc(x.shape, act_ld(x), ys, cout, _K1, _S1, _P0, (VS_CONV_STATS if stats else 0) | VS_CONV_NODEEP)
    dw = make_desc(x.shape, act_ld(x), ys, cout, _K1, _S1, _P0, VS_WGRAD_NODEEP)  # (the transform lives in the ring kernel)
    lib = _lib.load()
    return bool(lib.vs_conv_aol_ok(C.byref(d))) and bool(lib.vs_conv_wgrad_aol_ok(C.byref(dw)))


def conv_fwd_aol(x, w, in_scale, in_shift, out=None, stats=True):
    """conv1x1(relu(x * in_scale + in_shift), w): x is the producer unit's raw convolution output, the operand is the
    tensor bn_apply(x, in_scale, in_shift, relu=True) would have stored (same bits), formed on load.
    Returns (y, partials|None)."""
    cout = w.shape[0]
    ys = conv_out_shape(x.shape, cout, _K1, _S1, _P0)
    if out is None:
        out = new_act(*ys, device=x.device)
    elif tuple(out.shape) != ys:
        raise _lib.VsError(f"conv out shape {tuple(out.shape)} != {ys}")
    check_weight(w, cout, x.shape[1], _K1)
    # VS_CONV_NODEEP: the transform lives in the pointwise / 128 x 128 ring kernels; without the flag the rows query
    # below would answer for the deep-pipeline kernel's 256-row tiles
    d = make_desc(x.shape, act_ld(x), ys, act_ld(out), _K1, _S1, _P0, (VS_CONV_STATS if stats else 0) | VS_CONV_NODEEP)
    partials = None
    if stats:
        rows = _lib.load().vs_conv_stats_rows(C.byref(d))
        partials = torch.empty((rows, 2, cout), dtype=torch.float32, device=x.device)
    _lib.call("vs_conv_fwd_aol", _ptr(x), _ptr(w), _ptr(out), C.byref(d), _ptr(in_scale), _ptr(in_shift),
              _ptr(partials), _stream())
    return out, partials


def conv_wgrad_aol(dy, x, in_scale, in_shift, out=None):
    """dW of the convolution of conv_fwd_aol (x: the producer's raw output)."""
    cout, cin = dy.shape[1], x.shape[1]
    if out is None:
        out = torch.empty((cout, 1, 1, 1, cin), dtype=torch.float32, device=x.device).permute(0, 4, 1, 2, 3)
    elif not out.permute(0, 2, 3, 4, 1).is_contiguous() or out.dtype != torch.float32:
        raise _lib.VsError("conv_wgrad out must be fp32 with [Cout][taps][Cin] memory")
    d = make_desc(x.shape, act_ld(x), dy.shape, act_ld(dy), _K1, _S1, _P0, VS_WGRAD_NODEEP)  # ring kernel: it has the transform
    need = _lib.load().vs_conv_wgrad_workspace_bytes(C.byref(d))
    ws = _workspace(need, x.device, "wgrad") if need else None
    _lib.call("vs_conv_wgrad_aol", _ptr(dy), _ptr(x), _ptr(out), C.byref(d), _ptr(in_scale), _ptr(in_shift),
              _ptr(ws), C.c_size_t(ws.numel() if ws is not None else 0), _stream())
    return out


def _bc_desc(x, wb, k, s, p):
    ys = conv_out_shape(x.shape, wb.shape[0], k, s, p)
    return ys, make_desc(x.shape, act_ld(x), ys, wb.shape[0], k, s, p, VS_CONV_AFFINE | VS_CONV_RELU)


def conv_fwd_bc_fusable(x, wb, k, s, p, cout_c):
    """True if conv b (weights wb on input x) + a 1x1x1 conv c of cout_c channels run as one launch (eval)."""
    _, d = _bc_desc(x, wb, k, s, p)
    return bool(_lib.load().vs_conv_fwd_bc_fusable(C.byref(d), int(cout_c)))


def conv_fwd_bc(x, wb, k, s, p, scale_b, shift_b, wc, scale_c, shift_c, residual=None, relu=True, out=None):
    """relu?(conv1x1(relu(conv(x, wb) * scale_b + shift_b), wc) * scale_c + shift_c (+ residual)) in one launch
    (evaluation; the fast pathway's 8 / 16-channel bottlenecks)."""
    ys_b, d = _bc_desc(x, wb, k, s, p)
    cout_c = wc.shape[0]
    check_weight(wb, wb.shape[0], x.shape[1], k)
    check_weight(wc, cout_c, wb.shape[0], (1, 1, 1))
    ys = (ys_b[0], cout_c, *ys_b[2:])
    if out is None:
        out = new_act(*ys, device=x.device)
    elif tuple(out.shape) != ys:
        raise _lib.VsError(f"conv out shape {tuple(out.shape)} != {ys}")
    if residual is not None and tuple(residual.shape) != ys:
        raise _lib.VsError("residual shape mismatch")
    _lib.call("vs_conv_fwd_bc", _ptr(x), _ptr(wb), C.byref(d), _ptr(scale_b), _ptr(shift_b), _ptr(wc),
              C.c_int(cout_c), _ptr(scale_c), _ptr(shift_c), _ptr(residual),
              C.c_int(act_ld(residual) if residual is not None else 0), _ptr(out), C.c_int(act_ld(out)),
              C.c_int(1 if relu else 0), _stream())
    return out


def pack_stem_weight(w, out=None):
    """fp32 [Cout,3,kT,7,7] -> bf16 [ceil16(Cout)][kT][7][8][4] (zero padded) for the stem kernel."""
    cout, cin, kt, kh, kw = w.shape
    cp = (cout + 15) // 16 * 16
    if out is None:
        out = torch.zeros((cp, kt, 7, 8, 4), dtype=BF16, device=w.device)
    out[:cout, :, :, :7, :cin].copy_(w.detach().permute(0, 2, 3, 4, 1))
    return out


def stem_conv_fwd(x4, wp, cout, kt, out=None, scale=None, shift=None, relu=False, stats=False):
    """Conv3d(3->cout,[kt,7,7],s[1,2,2],p[kt//2,3,3]) on the C=4 packed input.
    Returns (y, partials|None)."""
    n, c, t, h, w = x4.shape
    if c != 4 or act_ld4(x4) != 4:
        raise _lib.VsError("stem input must be the dense C=4 packed activation")
    ho, wo = (h + 6 - 7) // 2 + 1, (w + 6 - 7) // 2 + 1
    if out is None:
        out = new_act(n, cout, t, ho, wo, x4.device)
    flags = (VS_CONV_AFFINE if scale is not None else 0) | (VS_CONV_RELU if relu else 0) | \
        (VS_CONV_STATS if stats else 0)
    partials = None
    if stats:
        rows = _lib.load().vs_stem_stats_rows(n, t, h, w)
        partials = torch.empty((rows, 2, cout), dtype=torch.float32, device=x4.device)
    _lib.call("vs_stem_conv_fwd", _ptr(x4), _ptr(wp), _ptr(out), n, t, h, w, cout, kt, act_ld(out),
              flags, _ptr(scale), _ptr(shift), _ptr(partials), _stream())
    return out, partials


def stem_conv_wgrad(dy, x4, kt):
    """Weight gradient of the stem conv; returns fp32 [Cout,3,kT,7,7]-shaped view of the padded
    [Cout][kT][7][8][4] result."""
    n, _, t, h, w = x4.shape
    cout = dy.shape[1]
    dwp = torch.empty((cout, kt, 7, 8, 4), dtype=torch.float32, device=dy.device)
    need = _lib.load().vs_stem_wgrad_workspace_bytes(n, t, h, w, cout, kt)
    ws = _workspace(need, dy.device)
    _lib.call("vs_stem_conv_wgrad", _ptr(dy), _ptr(x4), _ptr(dwp), n, t, h, w, cout, kt, act_ld(dy),
              _ptr(ws), C.c_size_t(ws.numel()), _stream())
    return dwp[:, :, :, :7, :3].permute(0, 4, 1, 2, 3)


def act_ld4(x):
    n, c, t, h, w = x.shape
    ok = x.dtype == BF16 and x.permute(0, 2, 3, 4, 1).is_contiguous()
    if not ok:
        raise _lib.VsError("not a dense channels-last activation")
    return c


def weight_transpose(w, out=None):
    """[Cout][taps][Cin] -> [Cin][taps][Cout] bf16 (logical [Cin,Cout,kT,kH,kW])."""
    cout, cin, kt, kh, kw = w.shape
    wt = out if out is not None else torch.empty(
        (cin, kt, kh, kw, cout), dtype=BF16, device=w.device).permute(0, 4, 1, 2, 3)
    _lib.call("vs_weight_transpose", _ptr(w), _ptr(wt), cout, kt * kh * kw, cin, _stream())
    return wt


_TILE_TABLES = {}


def _tile_table(table, ts):
    """(tile_first i64 [n] on the table's device, total tiles) of a batched-transpose table; cached."""
    key = (table.data_ptr(), ts)
    t = _TILE_TABLES.get(key)
    if t is None:
        rows = table.cpu().tolist()
        first, acc = [], 0
        for _, cout, taps, cin, _ in rows:
            first.append(acc)
            acc += taps * ((cout + ts - 1) // ts) * ((cin + ts - 1) // ts)
        t = (torch.tensor(first, dtype=torch.int64, device=table.device), acc, table)  # (keeps table alive)
        _TILE_TABLES[key] = t
    return t[0], t[1]


def weight_transpose_batched(src_arena, dst_arena, table, total, tiled=True):
    """All dgrad weight images [Cout][taps][Cin] -> [Cin][taps][Cout] (bf16) in one launch."""
    if not tiled:
        _lib.call("vs_weight_transpose_batched", _ptr(src_arena), _ptr(dst_arena), _ptr(table),
                  table.shape[0], int(total), _stream())
        return
    first, ntiles = _tile_table(table, 64)
    _lib.call("vs_weight_transpose_tiled", _ptr(src_arena), _ptr(dst_arena), _ptr(table), _ptr(first),
              table.shape[0], int(ntiles), 2, _stream())


def transpose_f32_batched(src, dst, table, total, tiled=True):
    """All fp32 linear weights [N][K] -> [K][N] in one launch."""
    if not tiled:
        _lib.call("vs_transpose_f32_batched", _ptr(src), _ptr(dst), _ptr(table), table.shape[0], int(total),
                  _stream())
        return
    first, ntiles = _tile_table(table, 32)
    _lib.call("vs_weight_transpose_tiled", _ptr(src), _ptr(dst), _ptr(table), _ptr(first), table.shape[0],
              int(ntiles), 4, _stream())


def conv_dgrad(dy, wt, xs, k, s, p, out=None, residual=None, naive=False, tile=None, ring=0,
               noclass=False, bn_stats=None, residual_bits=None, inplace=False, halo=True, pw=True,
               direct_bnb=False, bn_stats2=None, deep=True, splitk_il=None):
    """dx[xs] = conv_transpose(dy, w) (+ residual).  wt from weight_transpose.
    bn_stats = (y, mean, invstd, gamma, beta[, relu_bits]) of the BatchNorm + ReLU unit whose output this
    convolution consumed and whose complete dz this dx is: returns (dx, partial) with partial [rows, 2, Cin]
    the unit's BN-backward sums emitted by the dgrad epilogue (vs_conv_dgrad_bnstats), or (dx, None) when
    this dgrad cannot emit them.  Without `residual` the unit's mask is recomputed from gamma / beta; with
    `residual` the unit's bit mask (relu_bits, from bn_apply(want_bits=True)) is required.
    residual_bits: `residual` is an unmasked gradient and this the ReLU bit mask [rows, Cin/8] to apply to it.
    inplace: accumulate into `residual` (out = residual): a strided dgrad then only touches the positions its
    stride reaches."""
    if inplace:
        if residual is None or out is not None:
            raise _lib.VsError("conv_dgrad(inplace=True) accumulates into `residual`")
        out = residual
    if out is None:
        out = new_act(*xs, device=dy.device)
    flags = (VS_CONV_NAIVE if naive else 0) | (VS_CONV_RESIDUAL if residual is not None else 0)
    flags |= tile_flag("d", xs[0] * xs[2] * xs[3] * xs[4], xs[1], dy.shape[1] * k[0] * k[1] * k[2], k, s,
                       tile)
    flags |= (ring & 7) << 16  # VS_CONV_RING
    if noclass:
        flags |= 1 << 19  # VS_CONV_NOCLASS
    if not halo:
        flags |= 1 << 21  # VS_CONV_NOHALO
    elif halo == "force":
        flags |= 1 << 22  # VS_CONV_FORCEHALO
    if not pw:
        flags |= 1 << 23  # VS_CONV_NOPW
    elif pw == "force":
        flags |= 1 << 24  # VS_CONV_FORCEPW
    if direct_bnb:
        flags |= 1 << 25  # VS_CONV_DIRECTBNB
    if not deep:
        flags |= 1 << 27  # VS_CONV_NODEEP
    elif deep == "force":
        flags |= 1 << 28  # VS_CONV_FORCEDEEP
    if splitk_il is not None:
        flags |= (1 << 29) if splitk_il else (1 << 30)  # VS_CONV_SPLITK_IL / VS_CONV_NOSPLITK_IL
    two = bn_stats2 is not None and bn_stats is not None and residual is not None and tuple(s) == (1, 1, 1)
    if two:
        flags |= 1 << 26  # VS_CONV_BNB2
    d = make_desc(xs, act_ld(out), dy.shape, act_ld(dy), k, s, p, flags,
                  act_ld(residual) if residual is not None else 0)
    if residual_bits is not None:
        if residual is None or tuple(residual_bits.shape) != (xs[0] * xs[2] * xs[3] * xs[4], xs[1] // 8) \
                or residual_bits.dtype != torch.uint8:
            raise _lib.VsError("residual_bits must be uint8 [rows, Cin / 8] beside a residual")
    need = _lib.load().vs_conv_workspace_bytes(C.byref(d), 1)
    ws = _workspace(need, dy.device, "splitk") if need else None
    want_sums = bn_stats is not None
    ep = _lib.DgradEpilogue()
    ep.residual = residual.data_ptr() if residual is not None else None
    ep.residual_bits = residual_bits.data_ptr() if residual_bits is not None else None
    partial = partial2 = None
    if want_sums:
        y, mean, invstd, gamma, beta, bits = (tuple(bn_stats) + (None,))[:6]
        # pairings the kernel is built for: no residual + recomputed mask, residual + bit mask
        ok = (bits is not None) if residual is not None else (bits is None and gamma is not None)
        rows = _lib.load().vs_conv_dgrad_bnstats_rows(C.byref(d)) if ok else 0
        if rows > 0:
            partial = torch.empty((rows, 2, xs[1]), dtype=torch.float32, device=dy.device)
            ep.bn_y, ep.bn_y_ld = y.data_ptr(), act_ld(y)
            ep.relu_bits = bits.data_ptr() if bits is not None else None
            ep.mean, ep.invstd = mean.data_ptr(), invstd.data_ptr()
            ep.gamma = gamma.data_ptr() if gamma is not None else None
            ep.beta = beta.data_ptr() if beta is not None else None
            ep.stats_partial = partial.data_ptr()
            # a second unit fed by the same masked gradient (a ResBlock's shortcut unit beside its c unit):
            # bn_stats2 = (y2, mean2, invstd2); residual + bit-mask form, unit stride
            if two:
                y2, mean2, invstd2 = bn_stats2
                partial2 = torch.empty((rows, 2, xs[1]), dtype=torch.float32, device=dy.device)
                ep.bn_y2, ep.bn_y2_ld = y2.data_ptr(), act_ld(y2)
                ep.mean2, ep.invstd2 = mean2.data_ptr(), invstd2.data_ptr()
                ep.stats_partial2 = partial2.data_ptr()
    for t in (dy, wt, out, residual, residual_bits):
        _ptr(t)  # GPU-tensor check (the struct carries raw addresses)
    _lib.call("vs_conv_dgrad_ex", _ptr(dy), _ptr(wt), _ptr(out), C.byref(d), C.byref(ep), _ptr(ws),
              C.c_size_t(ws.numel() if ws is not None else 0), _stream())
    if bn_stats2 is not None:
        return out, partial, partial2
    return (out, partial) if want_sums else out


_ws_cache = {}
VS_WGRAD_NODEEP = 1 << 12  # include/vidsitu_hip.h: keep a weight gradient off the deep-pipeline kernel
import os as _os_wi
# VS_WHATIF (tools/whatif.sh: whole kernel families skipped to time what they cost -- GARBAGE numerics) is refused unless
# the tools-only guard VS_WHATIF_OK=1 stands beside it: a leaked variable must not train silently on garbage.
if int(_os_wi.environ.get("VS_WHATIF", "0")) != 0 and _os_wi.environ.get("VS_WHATIF_OK") != "1":
    raise RuntimeError("VS_WHATIF skips kernel launches (garbage numerics, timing experiments only): set VS_WHATIF_OK=1 "
                       "beside it, or unset it")
_WHATIF_WGRAD = (int(_os_wi.environ.get("VS_WHATIF", "0")) & 4) != 0  # tools only (see conv_wgrad)


# The slab reduce behind a weight gradient and the next unit's BN-backward finalize as ONE launch
# (vs_wgrad_reduce_defer): VS_REDUCE_MERGE=0 switches it off (A/B; the results are bitwise the same).
import os as _os_rm
REDUCE_MERGE = _os_rm.environ.get("VS_REDUCE_MERGE", "1") != "0"


def wgrad_reduce_defer(mode):
    _lib.call("vs_wgrad_reduce_defer", int(mode))


def wgrad_reduce_flush():
    _lib.call("vs_wgrad_reduce_flush")


def _workspace(nbytes, device, kind="scratch"):
    """Per-(device, stream) scratch.  kind = "wgrad": the weight-gradient slabs live in a buffer of their own -- a slab
    reduce may stay un-launched until the next BN-backward finalize on its stream (REDUCE_MERGE), and nothing else that
    takes a workspace (a split-K dgrad of the same unit, the stems' slabs, gemm_nt's partials) may write over slabs
    that are still waiting; only another weight gradient gets the same buffer, and vs_conv_wgrad flushes the pending
    reduce when it is handed the buffer it reads."""
    key = (device.index, torch.cuda.current_stream().cuda_stream, kind)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        # "splitk" (vs_conv_fwd / vs_conv_dgrad): the head of the buffer holds the arrival counters of the in-launch
        # split-K plan -- zero before the first launch, left at zero by every launch, written by nobody else
        # "fin" (bn_finalize / bn_bwd): arrival counters of the one-launch finalize -- zero before the first launch, left zero
        alloc = torch.zeros if kind in ("splitk", "fin") else torch.empty
        ws = alloc(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    return ws


class WgradBatch:
    """Weight-gradient slabs of many layers, reduced by ONE launch (`flush`) instead of one small launch behind
    every wgrad.  Each layer (keyed by its gradient's address) owns a persistent slab buffer -- 1.5 GB in all for a
    SlowFast-R50 step, on a 288 GB device -- so the set of (slab, dw) pairs is the same every step and the device
    table the batched kernel reads is built once (hipGraph-safe: nothing is allocated or uploaded in a replay)."""

    def __init__(self):
        self.slabs = {}      # dw address -> (uint8 slab tensor, nbytes)
        self.pending = []    # (slab address, dw address, elements, splits) since the last flush
        self.tables = {}     # tuple(pending) -> (device table, entries, total blocks)

    def wgrad(self, dy, x, k, s, p, out, ring=0):
        d = make_desc(x.shape, act_ld(x), dy.shape, act_ld(dy), k, s, p, (ring & 7) << 16)
        need = int(_lib.load().vs_conv_wgrad_workspace_bytes(C.byref(d)))
        key = out.data_ptr()
        slab = None
        if need:
            slab, have = self.slabs.get(key, (None, 0))
            if have < need:
                slab = torch.empty(need, dtype=torch.uint8, device=x.device)
                self.slabs[key] = (slab, need)
        splits = C.c_int(0)
        _lib.call("vs_conv_wgrad_partial", _ptr(dy), _ptr(x), _ptr(out), C.byref(d), _ptr(slab),
                  C.c_size_t(need), C.byref(splits), _stream())
        if splits.value > 1:
            self.pending.append((slab.data_ptr(), key, out.numel(), splits.value))
        return out

    def flush(self):
        """Sum every pending layer's slabs into its dw (one launch on the current stream)."""
        if not self.pending:
            return
        sig = tuple(self.pending)
        ent = self.tables.get(sig)
        if ent is None:
            rows, first = [], 0
            lib = _lib.load()
            for slab_ptr, dw_ptr, n, S in self.pending:
                rows.append([slab_ptr, dw_ptr, n, S, first])
                first += int(lib.vs_wgrad_reduce_blocks(n))
            dev = torch.device("cuda", torch.cuda.current_device())
            ent = self.tables[sig] = (torch.tensor(rows, dtype=torch.int64, device=dev), len(rows), first)
        table, n_ent, blocks = ent
        _lib.call("vs_wgrad_reduce_batched", _ptr(table), n_ent, blocks, _stream())
        self.pending = []


WG_TILES = [(128, 128), (128, 64), (64, 128), (64, 64), (32, 128), (32, 64), (16, 128), (16, 64)]


def conv_wgrad_split(dy, x, k, s, p, out, ring=0):
    """conv_wgrad as its two launches: runs the wgrad kernel now and returns the slab reduce as a callable (None when
    the plan has no position split and `out` is already complete).  The reduce reads only the stream's workspace and
    writes `out`, so a caller may release dy / x -- and signal whoever waits for them -- before it runs."""
    if not out.permute(0, 2, 3, 4, 1).is_contiguous() or out.dtype != torch.float32:
        raise _lib.VsError("conv_wgrad out must be fp32 with [Cout][taps][Cin] memory")
    d = make_desc(x.shape, act_ld(x), dy.shape, act_ld(dy), k, s, p, (ring & 7) << 16)
    need = int(_lib.load().vs_conv_wgrad_workspace_bytes(C.byref(d)))
    ws = _workspace(need, x.device, "wgrad") if need else None
    splits = C.c_int(0)
    _lib.call("vs_conv_wgrad_partial", _ptr(dy), _ptr(x), _ptr(out), C.byref(d), _ptr(ws), C.c_size_t(need),
              C.byref(splits), _stream())
    if splits.value <= 1:
        return None
    n, S = out.numel(), splits.value
    return lambda: _lib.call("vs_wgrad_reduce", _ptr(ws), _ptr(out), n, S, _stream())


WGRAD_GROUP_MAX = 20  # vs_conv_wgrad_group: items per launch


def _wgrad_items(items):
    arr = (_lib.WgradItem * len(items))()
    for i, (dy, x, k, s, p, out) in enumerate(items):
        if not out.permute(0, 2, 3, 4, 1).is_contiguous() or out.dtype != torch.float32:
            raise _lib.VsError("conv_wgrad out must be fp32 with [Cout][taps][Cin] memory")
        arr[i].dy, arr[i].x, arr[i].dw = dy.data_ptr(), x.data_ptr(), out.data_ptr()
        arr[i].d = make_desc(x.shape, act_ld(x), dy.shape, act_ld(dy), k, s, p, 0)
        _ptr(dy), _ptr(x), _ptr(out)
    return arr


def conv_wgrad_group_ok(items):
    """items: [(dy, x, k, s, p, dw_out)] -- can these weight gradients run as one grouped launch?"""
    if not 1 <= len(items) <= WGRAD_GROUP_MAX:
        return False
    return bool(_lib.load().vs_conv_wgrad_group_ok(_wgrad_items(items), len(items)))


def conv_wgrad_group(items):
    """The weight gradients of several convolutions as ONE deep-pipeline launch (+ one grouped slab reduce):
    vs_conv_wgrad_group.  items: [(dy, x, k, s, p, dw_out)], dw_out fp32 [Cout][taps][Cin], overwritten."""
    if _WHATIF_WGRAD:
        return
    arr = _wgrad_items(items)
    need = int(_lib.load().vs_conv_wgrad_group_workspace_bytes(arr, len(items)))
    ws = _workspace(need, items[0][0].device, "wgrad") if need else None
    _lib.call("vs_conv_wgrad_group", arr, len(items), _ptr(ws), C.c_size_t(ws.numel() if ws is not None else 0),
              _stream())


def conv_wgrad(dy, x, k, s, p, out=None, ring=0, batch=None, tile=None, slots=0, deep=True):
    """dw fp32, logical [Cout,Cin,kT,kH,kW], memory [Cout][taps][Cin].
    ring: 0 heuristic, 1 register-staged pipeline, 2 / 3 LDS-DMA ring stages (VS_CONV_RING).
    batch: a WgradBatch -- the split partials stay in the batch's slabs until `batch.flush()`."""
    cout, cin = dy.shape[1], x.shape[1]
    if out is None:
        out = torch.empty((cout, *k, cin), dtype=torch.float32, device=x.device).permute(0, 4, 1, 2, 3)
    elif not out.permute(0, 2, 3, 4, 1).is_contiguous() or out.dtype != torch.float32:
        raise _lib.VsError("conv_wgrad out must be fp32 with [Cout][taps][Cin] memory")
    if batch is not None:
        return batch.wgrad(dy, x, k, s, p, out, ring)
    if _WHATIF_WGRAD:  # timing experiment only (tools): what the step costs WITHOUT its weight gradients (garbage dW)
        return out
    # tile: index into WG_TILES, slots: block slots to fill (multiple of 8) -- tuning knobs, 0 / None = the plan
    flags = ((ring & 7) << 16) | (((tile + 1) << 8) if tile is not None else 0) | (((slots // 8) & 0xff) << 24)
    if not deep:
        flags |= 1 << 12  # VS_WGRAD_NODEEP: not the deep-pipeline kernel (A/B, tests)
    elif deep == "force":
        flags |= 1 << 13  # VS_WGRAD_FORCEDEEP
    d = make_desc(x.shape, act_ld(x), dy.shape, act_ld(dy), k, s, p, flags)
    need = _lib.load().vs_conv_wgrad_workspace_bytes(C.byref(d))
    ws = _workspace(need, x.device, "wgrad") if need else None
    _lib.call("vs_conv_wgrad", _ptr(dy), _ptr(x), _ptr(out), C.byref(d), _ptr(ws),
              C.c_size_t(ws.numel() if ws is not None else 0), _stream())
    return out


# ----------------------------------------------------------------------------
# batch norm
# ----------------------------------------------------------------------------
import os as _os
_WHATIF = int(_os.environ.get("VS_WHATIF", "0"))  # tools only: skip launches to measure what they cost on the step
_BN_TWO_LEVEL = int(_os.environ.get("VS_BN_TWO_LEVEL", "512"))  # partial rows above which a level-1 reduce runs first
# the BN finalizes (forward and backward) as ONE launch each whatever the number of partial rows (vs_bn_finalize_ws /
# vs_bn_bwd_finalize_ws); VS_BN_FIN2=0: the round-4 launches (A/B)
BN_FIN2 = _os.environ.get("VS_BN_FIN2", "1") != "0"
_fin_ws = [0]
_whatif_const = {}


def _fin_ws_bytes():
    if not _fin_ws[0]:
        _fin_ws[0] = int(_lib.load().vs_bn_finalize_workspace_bytes())
    return _fin_ws[0]



def bn_finalize(partials, count, gamma, beta, running_mean, running_var, momentum, eps, train):
    c = gamma.numel()
    dev = gamma.device
    scale = torch.empty(c, dtype=torch.float32, device=dev)
    shift = torch.empty(c, dtype=torch.float32, device=dev)
    mean = torch.empty(c, dtype=torch.float32, device=dev)
    invstd = torch.empty(c, dtype=torch.float32, device=dev)
    nparts = partials.shape[0] if train else 0
    if train and _WHATIF & 1:  # timing experiment only (garbage statistics): constants, no launch at all
        k = (c, str(dev))
        if k not in _whatif_const:
            _whatif_const[k] = (torch.ones(c, device=dev), torch.zeros(c, device=dev), torch.zeros(c, device=dev),
                                torch.ones(c, device=dev))
        return _whatif_const[k]
    if train and BN_FIN2:  # any number of rows in ONE launch (two levels inside it: vs_bn_finalize_ws)
        ws = _workspace(_fin_ws_bytes(), dev, "fin")
        _lib.call("vs_bn_finalize_ws", _ptr(partials), nparts, float(count), _ptr(gamma), _ptr(beta),
                  _ptr(running_mean), _ptr(running_var), float(momentum), float(eps), _ptr(scale), _ptr(shift),
                  _ptr(mean), _ptr(invstd), c, _ptr(ws), C.c_size_t(ws.numel()), _stream())
        return scale, shift, mean, invstd
    if train and nparts > _BN_TWO_LEVEL:  # two-level reduction keeps the finalize launch short
        lvl1 = torch.empty((32, 2, c), dtype=torch.float32, device=dev)
        _lib.call("vs_bn_partials_reduce", _ptr(partials), nparts, _ptr(lvl1), c, 32, _stream())
        partials, nparts = lvl1, 32
    _lib.call("vs_bn_finalize", _ptr(partials) if train else None, nparts, float(count),
              _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var), float(momentum),
              float(eps), _ptr(scale), _ptr(shift), _ptr(mean), _ptr(invstd), c, _stream())
    return scale, shift, mean, invstd


def bn_apply(y, scale, shift, residual=None, relu=True, out=None, want_bits=False):
    """out = relu?(y * scale + shift (+ residual)).  want_bits (with relu): also returns the ReLU
    mask as bits, uint8 [rows, C/8], for the backward passes (vs_bn_apply_mask)."""
    if out is None:
        out = new_act(*y.shape, device=y.device)
    if _WHATIF & 8:  # timing experiment only (tools): the step without its BN apply / backward passes (garbage tensors)
        if want_bits:
            return out, (torch.empty((act_rows(y), y.shape[1] // 8), dtype=torch.uint8, device=y.device) if relu else None)
        return out
    if want_bits and relu:
        bits = torch.empty((act_rows(y), y.shape[1] // 8), dtype=torch.uint8, device=y.device)
        _lib.call("vs_bn_apply_mask", _ptr(y), _ptr(scale), _ptr(shift), _ptr(residual), _ptr(out),
                  _ptr(bits), act_rows(y), y.shape[1], act_ld(y),
                  act_ld(residual) if residual is not None else 0, act_ld(out), _stream())
        return out, bits
    _lib.call("vs_bn_apply", _ptr(y), _ptr(scale), _ptr(shift), _ptr(residual), _ptr(out),
              act_rows(y), y.shape[1], act_ld(y), act_ld(residual) if residual is not None else 0,
              act_ld(out), int(relu), _stream())
    return (out, None) if want_bits else out


def bn_apply2(y, scale, shift, y2, scale2, shift2, out=None, want_bits=False):
    """out = relu(y * scale + shift + bf16(y2 * scale2 + shift2)): the c unit's apply pass with the shortcut unit's BN
    applied on the fly (vs_bn_apply2) -- bitwise bn_apply(y2, ...) then bn_apply(y, ..., residual); -> (out, bits | None)."""
    if out is None:
        out = new_act(*y.shape, device=y.device)
    bits = torch.empty((act_rows(y), y.shape[1] // 8), dtype=torch.uint8, device=y.device) if want_bits else None
    if _WHATIF & 8:
        return out, bits
    _lib.call("vs_bn_apply2", _ptr(y), _ptr(scale), _ptr(shift), _ptr(y2), _ptr(scale2), _ptr(shift2), _ptr(out),
              _ptr(bits), act_rows(y), y.shape[1], act_ld(y), act_ld(y2), act_ld(out), _stream())
    return out, bits


def bn_apply2_ok(c):
    cpr = c // 8
    return c % 8 == 0 and cpr & (cpr - 1) == 0


def bn_apply_maxpool(y, scale, shift, out=None, want_idx=True):
    """maxpool_hw(relu(y * scale + shift)) in one pass (the stems): -> (pooled, idx | None).  Bitwise bn_apply +
    maxpool_hw; the full-resolution normalised tensor is not written."""
    n, c, t, h, w = y.shape
    ho, wo = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    if out is None:
        out = new_act(n, c, t, ho, wo, y.device)
    idx = torch.empty((n, t, ho, wo, c), dtype=torch.uint8, device=y.device) if want_idx else None
    _lib.call("vs_bn_apply_maxpool", _ptr(y), _ptr(scale), _ptr(shift), _ptr(out), _ptr(idx), n, t, h, w, c,
              act_ld(y), act_ld(out), _stream())
    return out, idx


def bn_apply_maxpool_ok(y):
    n, c, t, h, w = y.shape
    cpr = c // 8
    return c % 8 == 0 and cpr & (cpr - 1) == 0 and n * t * h * w < (1 << 24)


def _bn_bwd_finalize(partial, nblk, dgamma, dbeta, c):
    if BN_FIN2:
        ws = _workspace(_fin_ws_bytes(), partial.device, "fin")
        _lib.call("vs_bn_bwd_finalize_ws", _ptr(partial), nblk, _ptr(dgamma), _ptr(dbeta), c, _ptr(ws),
                  C.c_size_t(ws.numel()), _stream())
    else:
        _lib.call("vs_bn_bwd_finalize", _ptr(partial), nblk, _ptr(dgamma), _ptr(dbeta), c, _stream())


def bn_bwd(dz, z, y, mean, invstd, gamma, relu, want_dres, dy_out=None, dgamma=None, dbeta=None,
           beta=None, zbits=None, partial=None, pool_src=None):
    """Returns (dy, dres|None, dgamma, dbeta); dgamma / dbeta may be given (param.grad views).
    ReLU mask source, in order of preference: `zbits` (uint8 [rows, C/8] from bn_apply), `z`
    (the unit's output), or -- with `beta` given and both None -- recomputed from y (units
    without a residual input).  `partial`: the sums already emitted by the dgrad that produced dz
    (conv_dgrad(bn_stats=...), recomputed-mask units only): the reduce pass is skipped."""
    rows, c = act_rows(y), y.shape[1]
    dev = y.device
    if pool_src is not None:
        # dz = maxpool_hw_bwd(d_pooled, idx), gathered inside the two passes (the stems: vs_bn_bwd_*_pool)
        if dz is not None or z is not None or zbits is not None or partial is not None or not relu or beta is None \
                or want_dres:
            raise _lib.VsError("bn_bwd(pool_src=...): a ReLU unit without residual input whose mask is recomputed")
        dp, pidx = pool_src
        n, _, t, h, w = y.shape
        nblk = _lib.load().vs_bn_bwd_reduce_rows(rows, c)
        partial = torch.empty((nblk, 2, c), dtype=torch.float32, device=dev)
        _lib.call("vs_bn_bwd_reduce_pool", _ptr(dp), _ptr(pidx), _ptr(y), _ptr(mean), _ptr(invstd), _ptr(gamma),
                  _ptr(beta), _ptr(partial), n, t, h, w, c, act_ld(dp), act_ld(y), _stream())
        dgamma = torch.empty(c, dtype=torch.float32, device=dev) if dgamma is None else dgamma
        dbeta = torch.empty(c, dtype=torch.float32, device=dev) if dbeta is None else dbeta
        _bn_bwd_finalize(partial, nblk, dgamma, dbeta, c)
        dy = new_act(*y.shape, device=dev) if dy_out is None else dy_out
        _lib.call("vs_bn_bwd_apply_pool", _ptr(dp), _ptr(pidx), _ptr(y), _ptr(mean), _ptr(invstd), _ptr(gamma),
                  _ptr(beta), _ptr(dgamma), _ptr(dbeta), _ptr(dy), n, t, h, w, c, act_ld(dp), act_ld(y), act_ld(dy),
                  _stream())
        return dy, None, dgamma, dbeta
    mode = int(relu)
    zz = z if relu else None
    z_ld = act_ld(zz) if zz is not None else 0
    if relu and zbits is not None:
        mode, zz, z_ld = 2, zbits, c // 8
    if partial is not None:
        if not (relu and ((zz is None and beta is not None) or zbits is not None)):
            raise _lib.VsError("bn_bwd: dgrad-emitted sums exist for recomputed-mask and bit-mask units only")
        nblk = partial.shape[0]
    else:
        nblk = _lib.load().vs_bn_bwd_reduce_rows(rows, c)
        if nblk <= 0:
            raise _lib.VsError("bn_bwd: unsupported channel count")
        partial = torch.empty((nblk, 2, c), dtype=torch.float32, device=dev)
        if not (_WHATIF & 8): _lib.call("vs_bn_bwd_reduce", _ptr(dz), _ptr(zz), _ptr(y), _ptr(mean), _ptr(invstd),
                  _ptr(gamma), _ptr(beta), _ptr(partial), rows, c, act_ld(dz), z_ld, act_ld(y),
                  mode, _stream())
    if dgamma is None:
        dgamma = torch.empty(c, dtype=torch.float32, device=dev)
    if dbeta is None:
        dbeta = torch.empty(c, dtype=torch.float32, device=dev)
    dy = new_act(*y.shape, device=dev) if dy_out is None else dy_out
    dres = new_act(*y.shape, device=dev) if want_dres else None
    if not (_WHATIF & 2):
        _bn_bwd_finalize(partial, nblk, dgamma, dbeta, c)
    if not (_WHATIF & 8): _lib.call("vs_bn_bwd_apply", _ptr(dz), _ptr(zz), _ptr(y), _ptr(mean), _ptr(invstd),
              _ptr(gamma), _ptr(beta), _ptr(dgamma), _ptr(dbeta), _ptr(dy), _ptr(dres), rows, c,
              act_ld(dz), z_ld, act_ld(y), act_ld(dy), act_ld(dres) if want_dres else 0,
              mode, _stream())
    return dy, dres, dgamma, dbeta


def bn_bwd_sums(dz, y, mean, invstd, zbits, dgamma, dbeta, partial=None):
    """The reduce + finalize half of bn_bwd for a unit whose gradient is dz under a ReLU bit mask: dgamma / dbeta are
    written; `partial`: the sums a data gradient already emitted (no reduce pass)."""
    rows, c = act_rows(y), y.shape[1]
    if partial is None:
        nblk = _lib.load().vs_bn_bwd_reduce_rows(rows, c)
        if nblk <= 0:
            raise _lib.VsError("bn_bwd: unsupported channel count")
        partial = torch.empty((nblk, 2, c), dtype=torch.float32, device=y.device)
        if not (_WHATIF & 8):
            _lib.call("vs_bn_bwd_reduce", _ptr(dz), _ptr(zbits), _ptr(y), _ptr(mean), _ptr(invstd), None, None,
                      _ptr(partial), rows, c, act_ld(dz), c // 8, act_ld(y), 2, _stream())
    if not (_WHATIF & 2):
        _bn_bwd_finalize(partial, partial.shape[0], dgamma, dbeta, c)


def bn_bwd_apply2(dz, zbits, a, b):
    """Backward apply of two units fed by the same masked gradient (vs_bn_bwd_apply2); a / b = (y, mean, invstd, gamma,
    dgamma, dbeta) -> (dy_a, dy_b).  Bitwise two bn_bwd apply passes."""
    ya, yb = a[0], b[0]
    rows, c = act_rows(ya), ya.shape[1]
    dya, dyb = new_act(*ya.shape, device=ya.device), new_act(*yb.shape, device=yb.device)
    if _WHATIF & 8:
        return dya, dyb
    _lib.call("vs_bn_bwd_apply2", _ptr(dz), _ptr(zbits), _ptr(ya), *[_ptr(t) for t in a[1:]], _ptr(dya),
              _ptr(yb), *[_ptr(t) for t in b[1:]], _ptr(dyb), rows, c, act_ld(dz), act_ld(ya), act_ld(dya),
              act_ld(yb), act_ld(dyb), _stream())
    return dya, dyb


def residual_add_f32(branch, residual, relu=True, out16=None):
    """fp32 residual stream (eval): -> (out16 bf16 activation, out32 fp32 [rows, C]).  `residual`: a bf16 activation
    (the shortcut unit's output) or the fp32 [rows, C] stream of the previous block."""
    rows, c = act_rows(branch), branch.shape[1]
    dev = branch.device
    if out16 is None:
        out16 = new_act(*branch.shape, device=dev)
    out32 = torch.empty((rows, c), dtype=torch.float32, device=dev)
    f32 = residual.dtype == torch.float32
    _lib.call("vs_residual_add_f32", _ptr(branch), _ptr(residual) if f32 else None, None if f32 else _ptr(residual),
              _ptr(out32), _ptr(out16), rows, c, act_ld(branch), c if f32 else act_ld(residual), c, act_ld(out16),
              int(relu), _stream())
    return out16, out32


# ----------------------------------------------------------------------------
# pooling
# ----------------------------------------------------------------------------
def maxpool_hw(x, out=None, want_idx=False):
    n, c, t, h, w = x.shape
    ho, wo = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    if out is None:
        out = new_act(n, c, t, ho, wo, x.device)
    idx = torch.empty((n, t, ho, wo, c), dtype=torch.uint8, device=x.device) if want_idx else None
    _lib.call("vs_maxpool_hw3s2_fwd", _ptr(x), _ptr(out), _ptr(idx), n, t, h, w, c, act_ld(x),
              act_ld(out), _stream())
    return out, idx


def maxpool_hw_bwd(dy, idx, xs):
    n, c, t, h, w = xs
    dx = new_act(n, c, t, h, w, dy.device)
    _lib.call("vs_maxpool_hw3s2_bwd", _ptr(dy), _ptr(idx), _ptr(dx), n, t, h, w, c, act_ld(dy),
              act_ld(dx), _stream())
    return dx


def maxpool_t(x, kt, want_idx=False):
    n, c, t, h, w = x.shape
    if act_ld(x) != c:
        raise _lib.VsError("maxpool_t needs a dense activation")
    out = new_act(n, c, t // kt, h, w, x.device)
    idx = torch.empty((n, t // kt, h, w, c), dtype=torch.uint8, device=x.device) if want_idx else None
    _lib.call("vs_maxpool_t_fwd", _ptr(x), _ptr(out), _ptr(idx), n, t, h * w, c, kt, _stream())
    return out, idx


def maxpool_t_bwd(dy, idx, xs, kt):
    n, c, t, h, w = xs
    dx = new_act(n, c, t, h, w, dy.device)
    _lib.call("vs_maxpool_t_bwd", _ptr(dy), _ptr(idx), _ptr(dx), n, t, h * w, c, kt, _stream())
    return dx


def avgpool_cat(feats):
    """AdaptiveAvgPool3d(1) per pathway + channel concat -> fp32 [N, sum C]."""
    n = feats[0].shape[0]
    ctot = sum(f.shape[1] for f in feats)
    out = torch.empty((n, ctot), dtype=torch.float32, device=feats[0].device)
    off = 0
    for f in feats:
        _, c, t, h, w = f.shape
        _lib.call("vs_avgpool_fwd", _ptr(f), _ptr(out), n, t * h * w, c, act_ld(f), ctot, off,
                  _stream())
        off += c
    return out


def avgpool_cat_bwd(dout, shapes):
    """dout fp32 [N, sum C] -> list of bf16 activations of `shapes`."""
    outs, off = [], 0
    dout = dout.contiguous()
    for (n, c, t, h, w) in shapes:
        dx = new_act(n, c, t, h, w, dout.device)
        _lib.call("vs_avgpool_bwd", _ptr(dout), _ptr(dx), n, t * h * w, c, act_ld(dx),
                  dout.shape[1], off, _stream())
        outs.append(dx)
        off += c
    return outs


# ----------------------------------------------------------------------------
# fp32 small ops (TxEncoder, heads, loss, optimizer)
# ----------------------------------------------------------------------------
def _f32c(t):
    if t.dtype != torch.float32:
        raise _lib.VsError("fp32 tensor expected")
    return t.contiguous()


def linear_fwd(x, w, b=None, relu=False):
    x, w = _f32c(x), _f32c(w)
    m, k = x.shape
    n = w.shape[0]
    y = torch.empty((m, n), dtype=torch.float32, device=x.device)
    _lib.call("vs_linear_fwd", _ptr(x), _ptr(w), _ptr(b), _ptr(y), m, n, k, int(relu), _stream())
    return y


_LINEAR_BWD_FUSED = _os.environ.get("VS_LINEAR_BWD_FUSED", "1") != "0"  # A/B: 0 = relu_bwd + bwd_data + bwd_weight launches


def linear_bwd(dy, x, w, need_dx=True, has_bias=True, dw_out=None, db_out=None, wt=None, relu_y=None, dx_res=None):
    """dw_out / db_out: write the parameter gradients in place (gradient-arena views).
    wt: an up-to-date [K][N] image of w (the parameter arena keeps one); else transposed here.
    relu_y: the layer's ReLU output -- dy is masked by (relu_y > 0) inside the kernels (no relu_bwd launch).
    dx_res [M, K]: added to dx (a gradient arriving over a residual connection around the layer)."""
    dy, x, w = _f32c(dy), _f32c(x), _f32c(w)
    m, n = dy.shape
    k = x.shape[1]
    dx = None
    if need_dx and _LINEAR_BWD_FUSED and m <= 8 and n % 4 == 0 and n <= 4096 and dy.data_ptr() % 16 == 0 \
            and (relu_y is None or relu_y.data_ptr() % 16 == 0):
        # few rows (the encoder / head section of the step): both gradients and the ReLU mask behind one launch
        if wt is None:
            wt = torch.empty((k, n), dtype=torch.float32, device=x.device)
            _lib.call("vs_transpose_f32", _ptr(w), _ptr(wt), n, k, _stream())
        if wt.data_ptr() % 16 == 0:
            dx = torch.empty((m, k), dtype=torch.float32, device=x.device)
            dw = dw_out if dw_out is not None else torch.empty((n, k), dtype=torch.float32, device=x.device)
            db = None
            if has_bias:
                db = db_out if db_out is not None else torch.empty(n, dtype=torch.float32, device=x.device)
            if dx_res is not None and (dx_res.dtype != torch.float32 or not dx_res.is_contiguous()):
                dx_res = _f32c(dx_res)
            _lib.call("vs_linear_bwd_fused_res", _ptr(dy), _ptr(relu_y), _ptr(x), _ptr(wt), _ptr(dx_res), _ptr(dx),
                      _ptr(dw), _ptr(db), m, n, k, _stream())
            return dx, dw, db
    if relu_y is not None:
        dy = relu_bwd(dy, relu_y)
    if need_dx:
        if wt is None:
            wt = torch.empty((k, n), dtype=torch.float32, device=x.device)
            _lib.call("vs_transpose_f32", _ptr(w), _ptr(wt), n, k, _stream())
        dx = torch.empty((m, k), dtype=torch.float32, device=x.device)
        _lib.call("vs_linear_bwd_data", _ptr(dy), _ptr(wt), _ptr(dx), m, n, k, _stream())
    dw = dw_out if dw_out is not None else torch.empty((n, k), dtype=torch.float32, device=x.device)
    db = None
    if has_bias:
        db = db_out if db_out is not None else torch.empty(n, dtype=torch.float32, device=x.device)
    _lib.call("vs_linear_bwd_weight", _ptr(dy), _ptr(x), _ptr(dw), _ptr(db), m, n, k, _stream())
    if dx_res is not None and dx is not None:
        dx = dx + dx_res.reshape(dx.shape)
    return dx, dw, db


_ones_cache = {}


class DropoutPool:
    """All dropout masks of one forward pass from ONE generator launch: the first pass records the
    (shape, p) sequence it is asked for; later passes with the same sequence draw a single flat
    mask of the total size (`F.dropout` on cached ones, hipGraph-safe) and hand out views.  A
    pass that asks for anything else falls back to one launch per mask."""

    def __init__(self):
        self.plan, self.seen, self.flat, self.pos, self.idx = None, [], None, 0, 0

    def begin(self, device):
        if self.seen and self.plan is None:
            ps = {p for _, p in self.seen}
            if len(ps) == 1:
                self.plan = list(self.seen)
        self.seen, self.idx, self.pos, self.flat = [], 0, 0, None
        if self.plan is not None:
            total = sum(int(torch.Size(sh).numel()) for sh, _ in self.plan)
            self.flat = dropout_mask((total,), self.plan[0][1], device)

    def get(self, shape, p, device):
        shape = tuple(shape)
        self.seen.append((shape, p))
        if self.flat is not None and self.idx < len(self.plan) and self.plan[self.idx] == (shape, p):
            n = int(torch.Size(shape).numel())
            out = self.flat[self.pos: self.pos + n].view(shape)
            self.pos += n
            self.idx += 1
            return out
        self.flat = None  # sequence changed: per-mask launches for the rest of this pass
        self.plan = None
        return dropout_mask(shape, p, device)


def dropout_mask(shape, p, device):
    """0 or 1/(1-p) per element, from torch's (hipGraph-safe) generator -- one launch."""
    key = (tuple(shape), str(device))
    ones = _ones_cache.get(key)
    if ones is None:
        ones = _ones_cache[key] = torch.ones(shape, dtype=torch.float32, device=device)
    return torch.nn.functional.dropout(ones, p, True)


def attn_small_fwd(q, k, v, n_heads, scale, drop_mask=None):
    q, k, v = _f32c(q), _f32c(k), _f32c(v)
    b, l, d = q.shape
    o = torch.empty_like(q)
    probs = torch.empty((b, n_heads, l, l), dtype=torch.float32, device=q.device)
    _lib.call("vs_attn_small_fwd", _ptr(q), _ptr(k), _ptr(v), _ptr(o), _ptr(probs),
              _ptr(drop_mask), b, l, n_heads, d // n_heads, 0, float(scale), _stream())
    return o, probs


def attn_small_bwd(q, k, v, probs, do, n_heads, scale, drop_mask=None):
    do = _f32c(do)
    b, l, d = q.shape
    dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
    _lib.call("vs_attn_small_bwd", _ptr(q), _ptr(k), _ptr(v), _ptr(probs), _ptr(do), _ptr(dq),
              _ptr(dk), _ptr(dv), _ptr(drop_mask), b, l, n_heads, d // n_heads, 0, float(scale),
              _stream())
    return dq, dk, dv


def attn_small_fwd_fused(qkv, b, l, n_heads, scale, drop_mask=None):
    """Self-attention on a fused projection buffer qkv f32 [b*l, 3d] (columns q | k | v) -> (o [b,l,d],
    probs): the kernel reads the three column blocks in place (row pitch 3d)."""
    d = qkv.shape[1] // 3
    o = torch.empty((b, l, d), dtype=torch.float32, device=qkv.device)
    probs = torch.empty((b, n_heads, l, l), dtype=torch.float32, device=qkv.device)
    _lib.call("vs_attn_small_fwd", _ptr(qkv), _ptr(qkv[:, d:]), _ptr(qkv[:, 2 * d:]), _ptr(o), _ptr(probs),
              _ptr(drop_mask), b, l, n_heads, d // n_heads, 3 * d, float(scale), _stream())
    return o, probs


def attn_small_bwd_fused(qkv, probs, do, b, l, n_heads, scale, drop_mask=None):
    """-> dqkv f32 [b*l, 3d] (dq | dk | dv written in place, row pitch 3d)."""
    do = _f32c(do)
    d = qkv.shape[1] // 3
    dqkv = torch.empty_like(qkv)
    _lib.call("vs_attn_small_bwd", _ptr(qkv), _ptr(qkv[:, d:]), _ptr(qkv[:, 2 * d:]), _ptr(probs), _ptr(do),
              _ptr(dqkv), _ptr(dqkv[:, d:]), _ptr(dqkv[:, 2 * d:]), _ptr(drop_mask), b, l, n_heads,
              d // n_heads, 3 * d, float(scale), _stream())
    return dqkv


class conv_pair:
    """`with conv_pair():` -- an eligible data gradient and weight gradient issued inside the block leave as ONE launch at
    its end (vs_conv_pair_begin / _end, csrc/conv_pair.hip); anything else is launched as usual."""

    def __enter__(self):
        _lib.call("vs_conv_pair_begin")
        return self

    def __exit__(self, *exc):
        _lib.call("vs_conv_pair_end")
        return False


def conv_pair_count():
    return int(_lib.load().vs_conv_pair_count())


def add_layernorm_fwd(x, r, gamma, beta, eps=1e-5, rmask=None):
    x = _f32c(x)
    r = _f32c(r) if r is not None else None
    rows, d = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    _lib.call("vs_add_layernorm_fwd", _ptr(x), _ptr(r), _ptr(rmask), _ptr(gamma), _ptr(beta),
              _ptr(y), _ptr(mean), _ptr(rstd), rows, d, float(eps), _stream())
    return y, mean, rstd


def add_layernorm_bwd(dy, x, r, gamma, mean, rstd, rmask=None, dg_out=None, db_out=None):
    """Returns (dx, dr, dgamma, dbeta); dr = dx * rmask.  dg_out / db_out: write the parameter
    gradients in place (gradient-arena views, overwrite semantics)."""
    dy = _f32c(dy)
    rows, d = x.shape
    dx = torch.empty_like(x)
    dr = torch.empty_like(x) if rmask is not None else None
    dg = torch.empty(d, dtype=torch.float32, device=x.device) if dg_out is None else dg_out
    db = torch.empty(d, dtype=torch.float32, device=x.device) if db_out is None else db_out
    _lib.call("vs_add_layernorm_bwd", _ptr(dy), _ptr(x), _ptr(r), _ptr(rmask), _ptr(gamma),
              _ptr(mean), _ptr(rstd), _ptr(dx), _ptr(dr), _ptr(dg), _ptr(db), rows, d, _stream())
    return dx, (dr if dr is not None else dx), dg, db


def softmax_xent(logits, labels, want_grad=True):
    logits = _f32c(logits)
    labels = labels.contiguous()
    rows, v = logits.shape
    loss = torch.empty((), dtype=torch.float32, device=logits.device)
    dlogits = torch.empty_like(logits) if want_grad else None
    _lib.call("vs_softmax_xent", _ptr(logits), _ptr(labels), _ptr(loss), _ptr(dlogits), rows, v,
              _stream())
    return loss, dlogits


def softmax_topk(logits, k=5):
    logits = _f32c(logits)
    rows, v = logits.shape
    probs = torch.empty((rows, k), dtype=torch.float32, device=logits.device)
    idx = torch.empty((rows, k), dtype=torch.int64, device=logits.device)
    _lib.call("vs_softmax_topk", _ptr(logits), _ptr(probs), _ptr(idx), rows, v, k, _stream())
    return probs, idx


def adam_step(p, g, m, v, lr, beta1, beta2, eps, step, grad_scale=1.0):
    _lib.call("vs_adam_step", _ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), float(lr),
              float(beta1), float(beta2), float(eps), int(step), float(grad_scale), _stream())


def adam_step_dev(p, g, m, v, lr, beta1, beta2, eps, step_counter, grad_scale=1.0):
    """Adam with the step count in device memory (int32 tensor, incremented here)."""
    _lib.call("vs_adam_step_dev", _ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), float(lr),
              float(beta1), float(beta2), float(eps), _ptr(step_counter), float(grad_scale), _stream())


def adam_step_dev_cast(p, g, m, v, p_bf16, lr, beta1, beta2, eps, step_counter, grad_scale=1.0):
    """adam_step_dev that also writes the bf16 copy of the updated parameters."""
    _lib.call("vs_adam_step_dev_cast", _ptr(p), _ptr(g), _ptr(m), _ptr(v), _ptr(p_bf16), p.numel(),
              float(lr), float(beta1), float(beta2), float(eps), _ptr(step_counter),
              float(grad_scale), _stream())


def adam_step_dev_cast_g16(p, g16, m, v, p_bf16, lr, beta1, beta2, eps, step_counter, grad_scale=1.0):
    """adam_step_dev_cast with the gradients read from a bf16 buffer (bf16 all-reduce payload)."""
    if g16.dtype != BF16:
        raise _lib.VsError("adam_step_dev_cast_g16: bf16 gradients expected")
    _lib.call("vs_adam_step_dev_cast_g16", _ptr(p), _ptr(g16), _ptr(m), _ptr(v), _ptr(p_bf16), p.numel(),
              float(lr), float(beta1), float(beta2), float(eps), _ptr(step_counter),
              float(grad_scale), _stream())


def adam_tick(step_counter):
    _lib.call("vs_adam_tick", _ptr(step_counter), _stream())


def adam_step_dev_range(p, g, m, v, p_bf16, lr, beta1, beta2, eps, step_counter, grad_scale=1.0):
    """The Adam update of one arena range (views of equal length) against an already ticked step count;
    g fp32 or bf16; p_bf16 None or the bf16 copy's view."""
    _lib.call("vs_adam_step_dev_range", _ptr(p), _ptr(g), int(g.dtype == BF16), _ptr(m), _ptr(v), _ptr(p_bf16),
              p.numel(), float(lr), float(beta1), float(beta2), float(eps), _ptr(step_counter),
              float(grad_scale), _stream())


def cast_bf16(src_f32, dst_bf16):
    _lib.call("vs_cast_f32_to_bf16", _ptr(src_f32), _ptr(dst_bf16), src_f32.numel(), _stream())


# ----------------------------------------------------------------------------
# GPT-2 decoder + beam-search scoring (csrc/gpt2_ops.hip)
# ----------------------------------------------------------------------------
ACT_NONE, ACT_RELU, ACT_GELU_NEW = 0, 1, 2


def gemm_nt(x, w, b=None, res=None, act=ACT_NONE, out=None):
    """act(x[M,K] @ w[N,K]^T + b) + res, fp32."""
    x, w = _f32c(x), _f32c(w)
    m, k = x.shape
    n = w.shape[0]
    if w.shape[1] != k:
        raise _lib.VsError(f"gemm_nt: K mismatch {tuple(x.shape)} x {tuple(w.shape)}")
    y = torch.empty((m, n), dtype=torch.float32, device=x.device) if out is None else out
    need = _lib.load().vs_gemm_nt_f32_workspace_bytes(m, n, k) if m > 64 else 0
    if need:
        ws = _workspace(need, x.device)
        _lib.call("vs_gemm_nt_f32_ws", _ptr(x), _ptr(w), _ptr(b), _ptr(res), _ptr(y), m, n, k, int(act),
                  _ptr(ws), C.c_size_t(ws.numel()), _stream())
    else:
        _lib.call("vs_gemm_nt_f32", _ptr(x), _ptr(w), _ptr(b), _ptr(res), _ptr(y), m, n, k, int(act),
                  _stream())
    return y


_RESIZE_TABLES = {}


def resize_tables(in_size, out_size, device):
    """(bounds i32 [out, 2], kk i32 [out, ksize]) of one axis on `device`, plus the host bounds
    (vs_resize_coeffs: Pillow's precompute_coeffs / normalize_coeffs_8bpc); cached per shape."""
    key = (int(in_size), int(out_size), str(device))
    t = _RESIZE_TABLES.get(key)
    if t is None:
        lib = _lib.load()
        ksize = lib.vs_resize_ksize(int(in_size), int(out_size))
        bounds = torch.zeros((out_size, 2), dtype=torch.int32)
        kk = torch.zeros((out_size, ksize), dtype=torch.int32)
        _lib.check(lib.vs_resize_coeffs(int(in_size), int(out_size), C.c_void_p(bounds.data_ptr()),
                                        C.c_void_p(kk.data_ptr())), "vs_resize_coeffs")
        t = (bounds.to(device), kk.to(device), bounds)
        _RESIZE_TABLES[key] = t
    return t


def resize_bicubic_u8(frames, out_h=224, out_w=224):
    """u8 [..., H0, W0, 3] -> u8 [..., out_h, out_w, 3]: PIL's `img.resize((out_w, out_h))` (bicubic) of
    `VsituDS.read_img`, bit-exact, on the GPU."""
    if frames.dtype != torch.uint8 or frames.shape[-1] != 3 or frames.dim() < 3:
        raise _lib.VsError("resize_bicubic_u8 expects uint8 [..., H, W, 3]")
    frames = frames.contiguous()
    h0, w0 = frames.shape[-3], frames.shape[-2]
    n = frames.numel() // (h0 * w0 * 3)
    dev = frames.device
    bh, kh, _ = resize_tables(w0, out_w, dev)
    bv, kv, bv_host = resize_tables(h0, out_h, dev)
    y0 = int(bv_host[0, 0])
    y1 = int(bv_host[-1, 0] + bv_host[-1, 1])
    dst = torch.empty(frames.shape[:-3] + (out_h, out_w, 3), dtype=torch.uint8, device=dev)
    tmp = torch.empty(n * (y1 - y0) * out_w * 3, dtype=torch.uint8, device=dev) \
        if (w0 != out_w and h0 != out_h) else None
    _ptr(frames)
    _lib.call("vs_resize_bicubic_u8", _ptr(frames), _ptr(dst), _ptr(tmp), n, h0, w0, out_h, out_w, _ptr(bh),
              _ptr(kh), kh.shape[1], _ptr(bv), _ptr(kv), kv.shape[1], y0, y1, _stream())
    return dst


def _pad16(n):
    return (n + 15) // 16 * 16


def pack_rows_f32(src):
    """Row-major f32 [R, K] (K % 16 == 0) -> fragment-major copy (flat, ceil16(R) * K floats) for
    `gemm_nt_packed` (vs_pack_rows_f32: 16 x 16 blocks in v_mfma_f32_16x16x4_f32 operand order)."""
    src = _f32c(src)
    r, k = src.shape
    dst = torch.empty(_pad16(r) * k, dtype=torch.float32, device=src.device)
    _lib.call("vs_pack_rows_f32", _ptr(src), _ptr(dst), r, k, _stream())
    return dst


def unpack_rows_f32(packed, r, k):
    """Inverse of pack_rows_f32 as torch ops (tests / debugging)."""
    return packed.view(_pad16(r) // 16, k // 16, 4, 16, 4).permute(0, 3, 1, 2, 4).reshape(_pad16(r), k)[:r]


def gemm_nt_packed(x_packed, w_packed, m, n, k, b=None, res=None, act=ACT_NONE, y_packed=False):
    """act(x @ w^T + b) + res for 1..64 rows on fragment-major operands; the output is row-major
    [m, n], or fragment-major (flat, ceil16(m) * n floats) when y_packed."""
    if x_packed.numel() < _pad16(m) * k or w_packed.numel() < _pad16(n) * k:
        raise _lib.VsError("gemm_nt_packed: packed operand too small for the given shape")
    if y_packed:
        y = torch.empty(_pad16(m) * n, dtype=torch.float32, device=x_packed.device)
    else:
        y = torch.empty((m, n), dtype=torch.float32, device=x_packed.device)
    _lib.call("vs_gemm_nt_f32_packed", _ptr(x_packed), _ptr(w_packed), _ptr(b), _ptr(res), _ptr(y), int(m),
              int(n), int(k), int(act), int(bool(y_packed)), _stream())
    return y


def layernorm_fwd_packed(x, gamma, beta, eps=1e-5):
    """LayerNorm of f32 [rows, D] written fragment-major (flat, ceil16(rows) * D floats)."""
    x = _f32c(x)
    rows, d = x.shape
    y = torch.empty(_pad16(rows) * d, dtype=torch.float32, device=x.device)
    _lib.call("vs_layernorm_fwd_packed", _ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), rows, d, float(eps),
              _stream())
    return y


def gpt2_embed(tokens, wte, wpe, pos0=0):
    """tokens i64 [R, L] -> f32 [R*L, D] = wte[tok] + wpe[pos0 + l]."""
    tokens = tokens.contiguous()
    r, l = tokens.shape
    if pos0 + l > wpe.shape[0]:
        raise _lib.VsError(f"gpt2_embed: position {pos0 + l} beyond n_positions {wpe.shape[0]}")
    out = torch.empty((r * l, wte.shape[1]), dtype=torch.float32, device=wte.device)
    _lib.call("vs_gpt2_embed", _ptr(tokens), _ptr(wte), _ptr(wpe), _ptr(out), r, l, wte.shape[1],
              int(pos0), wte.shape[0], _stream())
    return out


def attn_causal(qkv, key_mask, r, l, n_head):
    """qkv f32 [R*L, 3D]; key_mask uint8 [R, L] or None -> f32 [R*L, D]."""
    d = qkv.shape[1] // 3
    out = torch.empty((r * l, d), dtype=torch.float32, device=qkv.device)
    _lib.call("vs_attn_causal_fwd", _ptr(qkv), _ptr(key_mask), _ptr(out), r, l, n_head, d // n_head,
              _stream())
    return out


def attn_decode(qkv, kcache, vcache, key_mask, t, ancestry=None, out_packed=False):
    """qkv f32 [rows, 3D]; caches f32 [rows, H, Lmax, dh] (updated in place at position t);
    ancestry i32 [rows, Lmax] (optional): cache row that holds position j of row r;
    out_packed: fragment-major output (flat, ceil16(rows) * D floats) for gemm_nt_packed."""
    rows, h, lmax, dh = kcache.shape
    if ancestry is not None and (ancestry.dtype != torch.int32 or tuple(ancestry.shape) != (rows, lmax)
                                 or not ancestry.is_contiguous()):
        raise _lib.VsError("attn_decode: ancestry must be a contiguous int32 [rows, Lmax] tensor")
    if out_packed:
        out = torch.empty(_pad16(rows) * h * dh, dtype=torch.float32, device=qkv.device)
    else:
        out = torch.empty((rows, h * dh), dtype=torch.float32, device=qkv.device)
    _lib.call("vs_attn_decode", _ptr(qkv), _ptr(kcache), _ptr(vcache), _ptr(key_mask), _ptr(ancestry),
              _ptr(out), rows, h, dh, lmax, int(t), int(bool(out_packed)), _stream())
    return out


def kv_gather(src, dst, index, length):
    rows_out = index.numel()
    _, h, lmax, dh = src.shape
    _lib.call("vs_kv_gather", _ptr(src), _ptr(dst), _ptr(index.contiguous()), rows_out, h, dh, lmax,
              int(length), _stream())


def beam_topk(logits, cum, forced, k, pad, eos, unk, unk_penalty=0.0, temperature=1.0,
              eos_only=False, ban_eos=False):
    logits = _f32c(logits)
    rows, v = logits.shape
    val = torch.empty((rows, k), dtype=torch.float32, device=logits.device)
    idx = torch.empty((rows, k), dtype=torch.int64, device=logits.device)
    flags = (1 if eos_only else 0) | (2 if ban_eos else 0)
    ws = torch.empty(int(_lib.load().vs_beam_topk_workspace_bytes(rows, v, int(k))), dtype=torch.uint8,
                     device=logits.device)
    _lib.call("vs_beam_topk", _ptr(logits), _ptr(cum), _ptr(forced), _ptr(val), _ptr(idx), rows, v,
              int(k), int(pad), int(eos), int(unk), float(unk_penalty), float(temperature), flags,
              _ptr(ws), ws.numel(), _stream())
    return val, idx


def xent_ignore(logits, labels, ignore_index):
    """Mean CE over rows whose label != ignore_index -> (loss 0-dim, count)."""
    logits = _f32c(logits)
    rows, v = logits.shape
    nll = torch.empty(rows, dtype=torch.float32, device=logits.device)
    out = torch.empty(2, dtype=torch.float32, device=logits.device)
    _lib.call("vs_xent_ignore", _ptr(logits), _ptr(labels.contiguous()), _ptr(nll), _ptr(out), rows, v,
              v, int(ignore_index), _stream())
    return out[0], out  # (loss, [loss, count] device pair for xent_ignore_grad)


# ---- GPT-2 decoder backward ------------------------------------------------------------------
def transpose_f32(x):
    """[R, C] fp32 -> [C, R] (vs_transpose_f32)."""
    x = _f32c(x)
    r, c = x.shape
    out = torch.empty((c, r), dtype=torch.float32, device=x.device)
    _lib.call("vs_transpose_f32", _ptr(x), _ptr(out), r, c, _stream())
    return out


def gelu_new_fwd(x):
    x = _f32c(x)
    y = torch.empty_like(x)
    _lib.call("vs_gelu_new_fwd", _ptr(x), _ptr(y), x.numel(), _stream())
    return y


def gelu_new_bwd(dy, x_pre):
    dy, x_pre = _f32c(dy), _f32c(x_pre)
    dx = torch.empty_like(dy)
    _lib.call("vs_gelu_new_bwd", _ptr(dy), _ptr(x_pre), _ptr(dx), dy.numel(), _stream())
    return dx


def add_f32(a, b, out=None):
    a, b = _f32c(a), _f32c(b)
    out = torch.empty_like(a) if out is None else out
    _lib.call("vs_add_f32", _ptr(a), _ptr(b), _ptr(out), a.numel(), _stream())
    return out


def colsum_f32(x, out=None):
    x = _f32c(x)
    m, n = x.shape
    out = torch.empty(n, dtype=torch.float32, device=x.device) if out is None else out
    _lib.call("vs_colsum_f32", _ptr(x), _ptr(out), m, n, _stream())
    return out


def attn_causal_bwd(qkv, key_mask, dout, r, l, n_head):
    d = qkv.shape[1] // 3
    dqkv = torch.empty_like(qkv)
    need = _lib.load().vs_attn_causal_bwd_scratch_bytes(r, l, n_head)
    ws = _workspace(need, qkv.device)
    _lib.call("vs_attn_causal_bwd", _ptr(qkv), _ptr(key_mask), _ptr(_f32c(dout)), _ptr(dqkv), _ptr(ws),
              C.c_size_t(ws.numel()), r, l, n_head, d // n_head, _stream())
    return dqkv


def gpt2_embed_bwd(tokens, dh, dwte, dwpe, pos0=0):
    r, l = tokens.shape
    _lib.call("vs_gpt2_embed_bwd", _ptr(tokens.contiguous()), _ptr(_f32c(dh)), _ptr(dwte), _ptr(dwpe), r, l,
              dwte.shape[1], int(pos0), dwte.shape[0], _stream())


def xent_ignore_grad(logits, labels, loss_out, ignore_index, grad_scale=1.0):
    """grad_scale: a python number, or a 0-dim / 1-element fp32 device tensor (read by the kernel: no host sync)."""
    logits = _f32c(logits)
    rows, v = logits.shape
    dl = torch.empty_like(logits)
    if torch.is_tensor(grad_scale):
        gs = grad_scale.reshape(1).to(torch.float32).contiguous()
        _lib.call("vs_xent_ignore_grad_dev", _ptr(logits), _ptr(labels.contiguous()), _ptr(loss_out), _ptr(dl),
                  rows, v, v, int(ignore_index), _ptr(gs), _stream())
        return dl
    _lib.call("vs_xent_ignore_grad", _ptr(logits), _ptr(labels.contiguous()), _ptr(loss_out), _ptr(dl), rows,
              v, v, int(ignore_index), float(grad_scale), _stream())
    return dl


def beam_step(row_val, row_idx, tok_in, tok_out, sc_in, sc_out, ignore, finished, nfin, remaining,
              fin_tok, fin_score, fin_pos, fin_len, reorder, bsz, beam, k, vocab, step, max_len, eos,
              normalize, len_penalty, anc_in=None, anc_out=None):
    """One step of the device-side beam-search bookkeeping (vs_beam_step); all tensors on the GPU."""
    anc_ld = 0 if anc_in is None else anc_in.shape[1]
    _lib.call("vs_beam_step", _ptr(row_val), _ptr(row_idx), _ptr(tok_in), _ptr(tok_out), _ptr(sc_in),
              _ptr(sc_out), _ptr(ignore), _ptr(finished), _ptr(nfin), _ptr(remaining), _ptr(fin_tok),
              _ptr(fin_score), _ptr(fin_pos), _ptr(fin_len), _ptr(reorder), _ptr(anc_in), _ptr(anc_out),
              int(anc_ld), int(bsz), int(beam), int(k),
              int(vocab), int(step), int(max_len), int(eos), int(bool(normalize)), float(len_penalty),
              _stream())


# ---- fairseq TransformerDecoder pieces ---------------------------------------------------------
def embed_pos_fwd(tokens, emb, pos_table, pos_idx, scale):
    """scale * emb[tokens] + pos_table[pos_idx] -> f32 [tokens.numel(), D]."""
    tokens, pos_idx = tokens.contiguous(), pos_idx.contiguous()
    out = torch.empty((tokens.numel(), emb.shape[1]), dtype=torch.float32, device=emb.device)
    _lib.call("vs_embed_pos_fwd", _ptr(tokens), _ptr(emb), _ptr(pos_table), _ptr(pos_idx), _ptr(out),
              tokens.numel(), emb.shape[1], float(scale), _stream())
    return out


def embed_scatter_bwd(tokens, dx, demb, scale, pad):
    """demb (zero-filled) += scale * dx per non-padding token."""
    tokens = tokens.contiguous()
    _lib.call("vs_embed_scatter_bwd", _ptr(tokens), _ptr(_f32c(dx)), _ptr(demb), tokens.numel(), demb.shape[1],
              float(scale), int(pad), _stream())


def relu_bwd(dy, y):
    dy, y = _f32c(dy), _f32c(y)
    dx = torch.empty_like(dy)
    _lib.call("vs_relu_bwd", _ptr(dy), _ptr(y), _ptr(dx), dy.numel(), _stream())
    return dx


# ---- non-local block pieces ----------------------------------------------------------------------
def maxpool_hw2(x):
    """MaxPool3d([1,2,2], stride [1,2,2]) of a dense channels-last activation -> (y, idx u8)."""
    n, c, t, h, w = x.shape
    if act_ld(x) != c:
        raise _lib.VsError("maxpool_hw2 needs a dense activation")
    y = new_act(n, c, t, h // 2, w // 2, x.device)
    idx = torch.empty((n * t * (h // 2) * (w // 2), c), dtype=torch.uint8, device=x.device)
    _lib.call("vs_maxpool_hw2_fwd", _ptr(x), _ptr(y), _ptr(idx), n * t, h, w, c, _stream())
    return y, idx


def maxpool_hw2_bwd(dy, idx, xs):
    n, c, t, h, w = xs
    dx = new_act(n, c, t, h, w, dy.device)
    _lib.call("vs_maxpool_hw2_bwd", _ptr(dy), _ptr(idx), _ptr(dx), n * t, h, w, c, _stream())
    return dx


def softmax_rows_bf16(x, rows, p, out=None):
    """Row softmax of a dense bf16 [rows][p] matrix held in any tensor of rows * p elements."""
    out = x if out is None else out
    _lib.call("vs_softmax_rows_bf16", _ptr(x), _ptr(out), int(rows), int(p), _stream())
    return out


def softmax_rows_bwd_bf16(prob, dprob, rows, p, scale, out=None):
    out = dprob if out is None else out
    _lib.call("vs_softmax_rows_bwd_bf16", _ptr(prob), _ptr(dprob), _ptr(out), int(rows), int(p), float(scale),
              _stream())
    return out


def colsum_bf16(x, out=None):
    """Per-channel sum of a channels-last bf16 activation -> f32 [C] (conv bias gradient)."""
    n, c, t, h, w = x.shape
    out = torch.empty(c, dtype=torch.float32, device=x.device) if out is None else out
    ld = act_ld4(x) if c == 4 else act_ld(x)  # (the stems' packed C = 4 input)
    _lib.call("vs_colsum_bf16", _ptr(x), _ptr(out), n * t * h * w, c, ld, _stream())
    return out
