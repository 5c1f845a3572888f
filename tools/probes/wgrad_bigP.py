"""Is the 7-9 % weight-gradient difference of s2.pathway1_res0 at full resolution (tests/test_gpu_parity_full.py,
224^2 case) the kernel or the operands?  vs_conv_wgrad against fp32 torch on the SAME bf16 operands at that layer's
shape (1 clip: 32 x 56 x 56 positions, Cin = 8), with unstructured operands and with the structure of the real ones
(x >= 0 with a large mean, dy summing to ~0 per channel: a BN-backward output)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from vidsitu_amd import ops
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from gpu_utils import rb, to_act

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
for n, t, hw in ((1, 32, 56), (2, 32, 16), (8, 32, 56)):
    for cin, cout, k, p in ((8, 8, (3, 1, 1), (1, 0, 0)), (8, 32, (1, 1, 1), (0, 0, 0)), (32, 8, (3, 1, 1), (1, 0, 0))):
        for structured in (False, True):
            x = torch.randn(n, cin, t, hw, hw, generator=g)
            dy = torch.randn(n, cout, t, hw, hw, generator=g)
            if structured:
                x = x.abs() + 1.0
                dy = dy - dy.mean(dim=(0, 2, 3, 4), keepdim=True)
            x, dy = rb(x), rb(dy)
            w = torch.zeros(cout, cin, *k, requires_grad=True)
            y = F.conv3d(x.double(), w.double(), padding=p)
            ref = torch.autograd.grad(y, w, dy.double())[0].float()
            got = ops.conv_wgrad(to_act(dy, dev), to_act(x, dev), k, (1, 1, 1), p).float().cpu()
            got = got.permute(0, 1, 2, 3, 4)[:, :cin]
            err = float((got - ref).norm() / ref.norm())
            print(f"n{n} t{t} hw{hw} cin{cin} cout{cout} k{k} structured={structured}: rel_l2 {err:.3e}  |ref| {float(ref.norm()):.3e}")
