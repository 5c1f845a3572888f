"""How much of a few-row linear launch is the weight stream being cold?  A dependent chain of 48 linear_fwd launches in a
hipGraph, 8 rows, 512 -> 2048 (4 MB of fp32 weights each): all on ONE weight tensor (L2-hot), cycling through 48
different ones (192 MB: L2-cold, beyond the memory-side cache too), and the same with a 300 MB streaming read between
two replays (everything cold at the start).  usage: python tools/probes/linear_cold_weights.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from vidsitu_amd import ops


def main():
    dev = torch.device("cuda", 0)
    f32 = dict(dtype=torch.float32, device=dev)
    rows, d, n, reps = 8, 512, 2048, 48
    x = torch.randn(rows, d, **f32)
    ws = [torch.randn(n, d, **f32) for _ in range(reps)]
    b = torch.randn(n, **f32)
    big = torch.randn(75_000_000, **f32)

    def chain(weights):
        g = torch.cuda.CUDAGraph()
        ops.linear_fwd(x, weights[0], b, True)
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for i in range(reps):
                ops.linear_fwd(x, weights[i % len(weights)], b, True)
        return g

    def timed(g, flush):
        best = 1e9
        for _ in range(5):
            if flush:
                big.add_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
        return best

    g_hot, g_cyc = chain(ws[:1]), chain(ws)
    print(f"one weight tensor            {timed(g_hot, False):5.2f} us per launch")
    print(f"48 weight tensors (192 MB)   {timed(g_cyc, False):5.2f} us per launch")
    print(f"48 tensors, 300 MB flush     {timed(g_cyc, True):5.2f} us per launch")


if __name__ == "__main__":
    main()
