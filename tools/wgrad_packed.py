"""What do ALL weight gradients of one train step cost when nothing else is in their way?  Every conv layer's wgrad
(x its count in the net, stems excluded) launched back to back, round-robin over S streams, replayed from one hipGraph:
the kernels have no dependencies on each other, so ramps and tails overlap.  Compare with the sum of the isolated
launches (tools/fwd_layer_times.py wgrad) and with the step's time without weight gradients (VS_WHATIF=4).
usage: python tools/wgrad_packed.py [streams=1,2,3,4]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidsitu_amd import ops
from tools.layer_table import rows

dev = torch.device("cuda:0")
layers = []
for name, M, N, K, k, s, xin in rows(n=8):
    if "stem" in name:
        continue
    taps = k[0] * k[1] * k[2]
    cin = K // taps
    p = (k[0] // 2, k[1] // 2, k[2] // 2)
    pos_in = xin // cin
    t, h, w = [(t, hw, hw) for t in (8, 32) for hw in (56, 28, 14, 7) if 8 * t * hw * hw == pos_in][0]
    x = ops.new_act(8, cin, t, h, w, dev); x.normal_()
    ys = ops.conv_out_shape(x.shape, N, k, s, p)
    dy = ops.new_act(*ys, device=dev); dy.normal_()
    dw = torch.empty((N, *k, cin), dtype=torch.float32, device=dev).permute(0, 4, 1, 2, 3)
    layers.append((name, dy, x, k, s, p, dw))
print(len(layers), "weight gradients per step")
for ns in [int(a) for a in (sys.argv[1].split(",") if len(sys.argv) > 1 else "1,2,3,4")]:
    streams = [torch.cuda.Stream() for _ in range(ns)]

    def run():
        main = torch.cuda.current_stream()
        for st in streams:
            st.wait_stream(main)
        for i, (name, dy, x, k, s, p, dw) in enumerate(layers):
            with torch.cuda.stream(streams[i % ns]):
                ops.conv_wgrad(dy, x, k, s, p, out=dw)
        for st in streams:
            main.wait_stream(st)

    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        run(); run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    print(f"{ns} stream(s): all weight gradients of a step in {best:.3f} ms")
