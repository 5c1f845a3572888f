"""What running a unit's weight gradient beside its data gradient is worth, with the price of the graph edges taken out:
per slow-pathway shape, a replayed hipGraph of N x [dgrad ; wgrad] on one stream against N x [fork ; dgrad || wgrad ; join],
and the fork + join pair measured on the same box with spin kernels (tools/graph_edge_cost.py: ~17 us) subtracted --
an estimate of what one launch holding both kernels' blocks would take."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidsitu_amd import ops

dev = torch.device("cuda:0")
lane = torch.cuda.Stream()
N = 30
SHAPES = [
    ("s2.c  64->256   [1,1,1]", 64, 8, 56, 56, 256, (1, 1, 1), (0, 0, 0), 3),
    ("s3.b  128->128  [1,3,3]", 128, 8, 28, 28, 128, (1, 3, 3), (0, 1, 1), 4),
    ("s3.c  128->512  [1,1,1]", 128, 8, 28, 28, 512, (1, 1, 1), (0, 0, 0), 4),
    ("s4.a  1024->256 [3,1,1]", 1024, 8, 14, 14, 256, (3, 1, 1), (1, 0, 0), 5),
    ("s4.b  256->256  [1,3,3]", 256, 8, 14, 14, 256, (1, 3, 3), (0, 1, 1), 6),
    ("s4.c  256->1024 [1,1,1]", 256, 8, 14, 14, 1024, (1, 1, 1), (0, 0, 0), 6),
    ("s5.a  2048->512 [3,1,1]", 2048, 8, 7, 7, 512, (3, 1, 1), (1, 0, 0), 2),
    ("s5.b  512->512  [1,3,3]", 512, 8, 7, 7, 512, (1, 3, 3), (0, 1, 1), 3),
    ("s5.c  512->2048 [1,1,1]", 512, 8, 7, 7, 2048, (1, 1, 1), (0, 0, 0), 3),
]


def timed(build):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        build(1)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            build(N)
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / N)
    return best


EDGE = float(os.environ.get("EDGE_US", "17.0"))
tot = [0.0, 0.0, 0.0]
for name, cin, t, h, w, cout, k, p, cnt in SHAPES:
    s = (1, 1, 1)
    x = ops.new_act(8, cin, t, h, w, dev).normal_()
    dy = ops.new_act(*ops.conv_out_shape(x.shape, cout, k, s, p), device=dev).normal_()
    wk = torch.randn(cout, cin, *k) * 0.02  # logical [Cout,Cin,kT,kH,kW] -> kernel layout [Cout][taps][Cin] bf16
    wt = ops.weight_transpose(wk.permute(0, 2, 3, 4, 1).contiguous().to(ops.BF16).to(dev).permute(0, 4, 1, 2, 3))
    dw = torch.empty(cout, *k, cin, device=dev).permute(0, 4, 1, 2, 3)  # fp32, memory [Cout][taps][Cin]

    def seq(n):
        for _ in range(n):
            ops.conv_dgrad(dy, wt, tuple(x.shape), k, s, p)
            ops.conv_wgrad(dy, x, k, s, p, out=dw)

    def par(n):
        main = torch.cuda.current_stream()
        for _ in range(n):
            lane.wait_stream(main)
            with torch.cuda.stream(lane):
                ops.conv_wgrad(dy, x, k, s, p, out=dw)
            ops.conv_dgrad(dy, wt, tuple(x.shape), k, s, p)
            main.wait_stream(lane)
    a, b = timed(seq), timed(par)
    est = max(b - EDGE, 0.0)
    tot[0] += a * cnt; tot[1] += b * cnt; tot[2] += est * cnt
    print(f"{name}  x{cnt}: dgrad ; wgrad {a:6.1f} us | fork / join {b:6.1f} us | without the edge pair ~{est:6.1f} us")
print(f"sum x count: sequential {tot[0] / 1e3:.3f} ms, lanes {tot[1] / 1e3:.3f} ms, one launch (estimate) {tot[2] / 1e3:.3f} ms")
