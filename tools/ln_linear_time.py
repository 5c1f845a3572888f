"""LayerNorm in the prologue of its consumer (vs_ln_linear_fwd / vs_ln_bwd_linear_bwd) against the separate launches, at the
encoder's shapes (8 rows, d = 512), dependent chains of 12 in a hipGraph.  usage: python tools/ln_linear_time.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vidsitu_amd import ops


def graph_time(fn, reps=12):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


def main():
    dev = torch.device("cuda", 0)
    rows, d = 8, 512
    f32 = dict(dtype=torch.float32, device=dev)
    x, r = torch.randn(rows, d, **f32), torch.randn(rows, d, **f32)
    rmask = (torch.rand(rows, d, **f32) > 0.1).float() / 0.9
    gamma, beta = torch.rand(d, **f32) + 0.5, torch.randn(d, **f32)
    y_ln, mean, rstd = torch.empty(rows, d, **f32), torch.empty(rows, **f32), torch.empty(rows, **f32)
    for name, n in (("q|k|v 512 -> 1536", 1536), ("ffn1 512 -> 2048", 2048)):
        w, b = torch.randn(n, d, **f32), torch.randn(n, **f32)

        def sep():
            yl, _, _ = ops.add_layernorm_fwd(x, r, gamma, beta, 1e-5, rmask)
            ops.linear_fwd(yl, w, b, True)

        def fused():
            ops.ln_linear_fwd(x, r, gamma, beta, 1e-5, rmask, y_ln, mean, rstd, w, b, True)

        t_ln = graph_time(lambda: ops.add_layernorm_fwd(x, r, gamma, beta, 1e-5, rmask))
        t_lin = graph_time(lambda: ops.linear_fwd(x, w, b, True))
        print(f"fwd {name:18s} LN {t_ln:5.1f}  linear {t_lin:5.1f}  LN;linear {graph_time(sep):5.1f}  fused {graph_time(fused):5.1f} us")
    ops.add_layernorm_fwd(x, r, gamma, beta, 1e-5, rmask)
    _, mean, rstd = ops.add_layernorm_fwd(x, r, gamma, beta, 1e-5, rmask)
    dy = torch.randn(rows, d, **f32)
    dg, db_ = torch.empty(d, **f32), torch.empty(d, **f32)
    dx_ln = torch.empty(rows, d, **f32)
    for name, k in (("wo 512 <- 512", 512), ("ffn2 512 <- 2048", 2048)):
        w = torch.randn(d, k, **f32)
        wt = w.t().contiguous()
        x_lin = torch.randn(rows, k, **f32)
        dw, dbl = torch.empty(d, k, **f32), torch.empty(d, **f32)

        def sep():
            _, dr, _, _ = ops.add_layernorm_bwd(dy, x, r, gamma, mean, rstd, rmask, dg_out=dg, db_out=db_)
            ops.linear_bwd(dr, x_lin, w, need_dx=True, has_bias=True, dw_out=dw, db_out=dbl, wt=wt)

        def fused():
            ops.ln_bwd_linear_bwd(dy, x, r, gamma, mean, rstd, rmask, dx_ln, dg, db_, x_lin, wt, dw, dbl)

        t_ln = graph_time(lambda: ops.add_layernorm_bwd(dy, x, r, gamma, mean, rstd, rmask, dg_out=dg, db_out=db_))
        t_lin = graph_time(lambda: ops.linear_bwd(dy, x_lin, w, need_dx=True, has_bias=True, dw_out=dw, db_out=dbl, wt=wt))
        print(f"bwd {name:18s} LN {t_ln:5.1f}  linear {t_lin:5.1f}  LN;linear {graph_time(sep):5.1f}  fused {graph_time(fused):5.1f} us")


if __name__ == "__main__":
    main()
