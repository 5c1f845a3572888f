"""Anatomy of a convolution launch at 8 clips: what the chain's launches cost beyond their bytes.

Two runs of this script, one box:
  VS_LIB_PATH=tmp/stamp/libvidsitu_hip.so python tools/launch_anatomy.py          (diagnostic build, -DVS_STAMP)
  python tools/launch_anatomy.py --plain                                          (the shipped library: times only)

The diagnostic build (tools/build_stamp_lib.sh) stamps every block of conv_igemm_kernel (register-staged and ring),
conv_halo_kernel and conv_pw_kernel with s_memrealtime (100 MHz, chip-wide) at: 0 block start, 1 tables / descriptors
ready, 2 first tile landed in LDS, 3 main loop done, 4 epilogue staged, 5 last store issued, 6 block end (conv_tile.h).
Each of the most frequent slow-pathway shapes of the 8-clip training step runs alone, replayed from a hipGraph
(REPS launches back to back; the stamps of the last one are read): forward with the training-mode epilogue (raw output +
BN batch-statistic partials) and the data gradient.  Printed per launch: event-timed us per launch, the grid and the
kernel plan, the span first block start -> last block end, when the median / last block STARTS (dispatch ramp and
residency rounds), and the median block's phases; then the first-starting and the last-ending block.
Never quote the stamped build's run time as the kernel's: read its SHARES."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tools.layer_table import rows
from vidsitu_amd import _lib, ops

dev = torch.device("cuda:0")
PLAIN = "--plain" in sys.argv
NCLIPS = int(next((a.split("=")[1] for a in sys.argv if a.startswith("--clips=")), 8))
TOP = int(next((a.split("=")[1] for a in sys.argv if a.startswith("--top=")), 12))
ONLY = next((a.split("=")[1].split(",") for a in sys.argv if a.startswith("--only=")), None)
REPS = 20
MAXB = 1 << 17  # blocks the stamp buffer holds

lib = _lib.load()
stamps = None
if not PLAIN:
    if not hasattr(lib, "vs_stamp_attach"):
        sys.exit("this library has no stamps: build tools/build_stamp_lib.sh and set VS_LIB_PATH (or pass --plain)")
    stamps = torch.zeros((MAXB, 8), dtype=torch.int64, device=dev)
    lib.vs_stamp_attach.argtypes = [C.c_void_p]
    lib.vs_stamp_attach(C.c_void_p(stamps.data_ptr()))

PLAN_NAMES = {0: "tile", 1: "direct", 2: "halo", 3: "pw", 4: "deep", 5: "tile+splitK"}


def plan(x, cout, k, s, p, y, dgrad, flags=0):
    d = ops.make_desc(x.shape, ops.act_ld(x), y.shape, ops.act_ld(y), k, s, p, flags)
    out = (C.c_int * 5)()
    lib.vs_conv_plan(C.byref(d), int(dgrad), C.cast(out, C.c_void_p))
    kind = PLAN_NAMES.get(out[4], str(out[4]))
    return f"{kind} {out[0]}x{out[1]} " + (f"ring{out[2]}" if kind.startswith("tile") and out[2] else
                                           ("reg" if kind.startswith("tile") else f"d{out[2]}")) + \
        (f" S{out[3]}" if kind.startswith("tile") and out[3] > 1 else "")


def graph_time(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REPS):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / REPS)
    if stamps is not None:  # one more replay on a clean buffer: the stamps read are the last launch's
        stamps.zero_()
        g.replay()
        torch.cuda.synchronize()
    return best * 1e3  # us


def anatomy():
    st = stamps.cpu().numpy()
    st = st[st[:, 0] > 0]
    if len(st) == 0:
        return None
    import numpy as np

    t = st[:, :7].astype(np.float64) * 0.01  # us (10 ns ticks)
    t0 = t[:, 0].min()
    span = t[:, 6].max() - t0
    starts = np.sort(t[:, 0] - t0)
    ph = np.diff(t, axis=1)  # [blocks, 6]: tables, first tile, loop, staging, stores, tail
    med = np.median(ph, axis=0)
    first = ph[np.argmin(t[:, 0])]
    last = ph[np.argmax(t[:, 6])]
    life = np.median(t[:, 6] - t[:, 0])
    return dict(blocks=len(st), span=span, start_med=float(np.median(starts)), start_last=float(starts[-1]),
                med=med, first=first, last=last, life=life, ksteps=int(np.median(st[:, 7])),
                last_start=float(t[np.argmax(t[:, 6]), 0] - t0))


agg = {}
for name, M, N, K, k, s, xin in rows(n=NCLIPS):
    if ".p0." not in name and not ONLY:
        continue
    a = agg.setdefault((M, N, K, k, s), [0, name, xin])
    a[0] += 1
shapes = sorted(agg.items(), key=lambda kv: (-kv[1][0], -kv[0][0] * kv[0][1] * kv[0][2]))
if ONLY:
    shapes = [kv for kv in shapes if any(o in kv[1][1] for o in ONLY)]
else:
    shapes = shapes[:TOP]

print(f"# tools/launch_anatomy.py, {NCLIPS} clips, {'shipped library (times only)' if PLAIN else 'DIAGNOSTIC build with stamps (read shares, not lengths)'}"
      f", lib {os.path.basename(os.path.dirname(_lib.LIB_PATH))}/{os.path.basename(_lib.LIB_PATH)}")
hdr = (f"{'layer':12s} {'x':>2s} {'dir':5s} {'M':>7s} {'N':>5s} {'K':>5s} {'plan':18s} {'us':>6s} {'MB':>6s} {'us@4.4TB/s':>10s}")
if not PLAIN:
    hdr += (f" | {'blocks':>6s} {'ksteps':>6s} {'span':>6s} {'start med/last':>14s} {'life':>5s} |"
            f" median block: {'tables':>6s} {'1st tile':>8s} {'loop':>6s} {'stage':>6s} {'stores':>6s} {'tail':>5s}")
print(hdr)
tot = {}
for (M, N, K, k, s), (cnt, name, xin) in shapes:
    taps = k[0] * k[1] * k[2]
    cin = K // taps
    p = (k[0] // 2, k[1] // 2, k[2] // 2)
    pos_in = xin // cin
    cands = [(t, hw, hw) for t in (8, 32) for hw in (56, 28, 14, 7) if NCLIPS * t * hw * hw == pos_in]
    t, h, w = cands[0]
    x = ops.new_act(NCLIPS, cin, t, h, w, dev); x.normal_()
    wt = (torch.randn(N, *k, cin, device=dev) / K ** 0.5).to(ops.BF16).permute(0, 4, 1, 2, 3)
    ys = ops.conv_out_shape(x.shape, N, k, s, p)
    assert ys[0] * ys[2] * ys[3] * ys[4] == M, (name, ys, M)
    by = 2.0 * (xin + M * N + N * K)
    out = ops.new_act(*ys, device=dev)
    dy = ops.new_act(*ys, device=dev); dy.normal_()
    wtt = ops.weight_transpose(wt)
    dx = ops.new_act(*x.shape, device=dev)
    for kd in ("fwd", "dgrad"):
        if kd == "fwd":
            fn = lambda: ops.conv_fwd(x, wt, k, s, p, out=out, stats=True)
            pl = plan(x, N, k, s, p, out, False, _lib.VS_CONV_STATS)
        else:
            fn = lambda: ops.conv_dgrad(dy, wtt, tuple(x.shape), k, s, p, out=dx)
            pl = plan(x, N, k, s, p, out, True)
        us = graph_time(fn)
        row = f"{name:12s} {cnt:2d} {kd:5s} {M:7d} {N:5d} {K:5d} {pl:18s} {us:6.1f} {by / 1e6:6.1f} {by / 4.4e6:10.1f}"
        tot.setdefault(kd, [0.0, 0.0])
        tot[kd][0] += us * cnt
        tot[kd][1] += by / 4.4e6 * cnt
        if not PLAIN:
            a = anatomy()
            if a is None:
                row += " | (no stamps: this plan's kernel is not instrumented)"
            else:
                f6 = lambda v: " ".join(f"{x_:6.2f}" for x_ in v)
                row += (f" | {a['blocks']:6d} {a['ksteps']:6d} {a['span']:6.1f} {a['start_med']:6.1f}/{a['start_last']:6.1f}  {a['life']:5.1f} |"
                        f"               {f6(a['med'])}")
                row += (f"\n{'':100s} first-starting block: {f6(a['first'])}\n{'':100s} last-ending block (started at {a['last_start']:5.1f}): {f6(a['last'])}")
        print(row, flush=True)
for kd, (us, fl) in tot.items():
    print(f"# {kd}: sum over these shapes x count {us / 1e3:.3f} ms; their bytes at 4.4 TB/s {fl / 1e3:.3f} ms")
