"""Stage-by-stage check of ops.TxStack against the stand-alone ops (bitwise)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidsitu_amd import ops

dev = torch.device("cuda", 0)
torch.manual_seed(0)
rows, d = 8, 1024
f = dict(device=dev, dtype=torch.float32)
a, b2 = torch.randn(rows, d, **f), torch.randn(rows, d, **f)
x, r = torch.randn(rows, d, **f), torch.randn(rows, d, **f)
rmask = torch.nn.functional.dropout(torch.ones(rows, d, **f), 0.1, True)
gamma, beta = torch.randn(d, **f), torch.randn(d, **f)
y, mean, rstd = ops.add_layernorm_fwd(x, r, gamma, beta, 1e-5, rmask)
# stand-alone: dy = a + b
dy = a + b2
dx0, dr0, dg0, db0 = ops.add_layernorm_bwd(dy, x, r, gamma, mean, rstd, rmask)
for two in (False, True):
    st = ops.TxStack(dev)
    dx1, dr1 = torch.empty_like(x), torch.empty_like(x)
    dg1, db1 = torch.empty_like(gamma), torch.empty_like(gamma)
    if two:
        st.add_layernorm_bwd(a, b2, x, r, rmask, gamma, mean, rstd, dx1, dr1, dg1, db1)
    else:
        st.add_layernorm_bwd(dy, None, x, r, rmask, gamma, mean, rstd, dx1, dr1, dg1, db1)
    st.run()
    torch.cuda.synchronize()
    print("LN_BWD two addends" if two else "LN_BWD one addend", [bool(torch.equal(p, q)) for p, q in ((dx0, dx1), (dr0, dr1), (dg0, dg1), (db0, db1))],
          "failed" if st.failed() else "")
# linear bwd, N = 3072 inner
n, k = 3072, 1024
dyq, xin = torch.randn(rows, n, **f), torch.randn(rows, k, **f)
w = torch.randn(n, k, **f) * 0.02
wt = w.t().contiguous()
dxa, dwa, _ = ops.linear_bwd(dyq, xin, w, need_dx=True, has_bias=False, wt=wt)
st = ops.TxStack(dev)
dxb, dwb = torch.empty_like(dxa), torch.empty_like(dwa)
st.linear_bwd(dyq, None, xin, wt, dxb, dwb, None)
st.run(); torch.cuda.synchronize()
print("LINBWD 3072", bool(torch.equal(dxa, dxb)), bool(torch.equal(dwa, dwb)))
# chain: LINBWD then LN_BWD reading its output (visibility across the barrier)
st = ops.TxStack(dev)
dxb2 = torch.empty_like(dxa)
dx2, dr2, dg2, db2 = torch.empty_like(x), torch.empty_like(x), torch.empty_like(gamma), torch.empty_like(gamma)
st.linear_bwd(dyq, None, xin, wt, dxb2, dwb, None)
st.add_layernorm_bwd(a, dxb2, x, r, rmask, gamma, mean, rstd, dx2, dr2, dg2, db2)
st.run(); torch.cuda.synchronize()
dxr, drr, dgr, dbr = ops.add_layernorm_bwd(a + dxa, x, r, gamma, mean, rstd, rmask)
print("chain", [bool(torch.equal(p, q)) for p, q in ((dxr, dx2), (drr, dr2), (dgr, dg2), (dbr, db2))], float((dxr - dx2).abs().max()))
