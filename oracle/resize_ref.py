"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of `PIL.Image.resize((w, h))` with its default filter (BICUBIC since Pillow 7.0) on
8-bit RGB images, i.e. of the step the reference's loader applies to every decoded frame:
`VsituDS.read_img`, `vidsitu_code/dat_loader.py:183-191` (`img.resize((224, 224))`).  The algorithm
lives in a third-party dependency that is absent from the reference tree: Pillow, pinned at
`pillow=7.2.0` (`vsitu_pyt_env.yml:151`), file `src/libImaging/Resample.c` -- two passes (horizontal,
then vertical over the rows the vertical pass needs), per output pixel a window of
`(int)(center - support + 0.5) .. (int)(center + support + 0.5)` source pixels, bicubic
(a = -0.5) weights evaluated in double precision at `(x - center + 0.5) / filterscale`, normalised,
rounded to 22-bit fixed point, accumulated from `1 << 21`, shifted and clipped to 0..255 after EACH pass.

Pinned: `tests/golden/resize_u8.npz` holds outputs of the installed Pillow (generating script
`tests/golden/gen_resize_golden.py`); `tests/test_oracle_resize.py` also compares against Pillow live
when it is importable.  Bit-exact.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
BICUBIC_SUPPORT = 2.0


def bicubic_filter(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size, out_size):
    """Resample.c `precompute_coeffs` + `normalize_coeffs_8bpc` for the whole-image box.
    -> ksize, bounds int32 [out, 2] (first source index, count), kk int32 [out, ksize]."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = BICUBIC_SUPPORT * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [bicubic_filter((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            f = v * (1 << PRECISION_BITS)
            kk[xx, x] = int(-0.5 + f) if v < 0 else int(0.5 + f)
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def _pass(img, bounds, kk, axis):
    """One resampling pass along `axis` (1 = horizontal, 0 = vertical) of u8 [H, W, C]."""
    src = img.astype(np.int64)
    n_out = bounds.shape[0]
    shape = list(img.shape)
    shape[axis] = n_out
    out = np.zeros(shape, dtype=np.uint8)
    for o in range(n_out):
        lo, cnt = int(bounds[o, 0]), int(bounds[o, 1])
        k = kk[o, :cnt].astype(np.int64)
        if axis == 1:
            acc = (src[:, lo:lo + cnt, :] * k[None, :, None]).sum(axis=1)
        else:
            acc = (src[lo:lo + cnt, :, :] * k[:, None, None]).sum(axis=0)
        acc = (acc + (1 << (PRECISION_BITS - 1))) >> PRECISION_BITS
        val = np.clip(acc, 0, 255).astype(np.uint8)
        if axis == 1:
            out[:, o, :] = val
        else:
            out[o, :, :] = val
    return out


def resize_bicubic_u8(img, out_h, out_w):
    """u8 [H, W, C] -> u8 [out_h, out_w, C] as `Image.fromarray(img).resize((out_w, out_h))`."""
    h, w = img.shape[:2]
    _, bh, kh = precompute_coeffs(w, out_w)
    _, bv, kv = precompute_coeffs(h, out_h)
    if w != out_w:  # Resample.c skips a pass whose size does not change (need_horizontal / _vertical)
        y0 = int(bv[0, 0]) if h != out_h else 0
        y1 = int(bv[-1, 0] + bv[-1, 1]) if h != out_h else h
        tmp = _pass(img[y0:y1], bh, kh, axis=1)
        if h != out_h:
            bv = bv.copy()
            bv[:, 0] -= y0
    else:
        tmp = img
    if h != out_h:
        tmp = _pass(tmp, bv, kv, axis=0)
    return tmp
