"""The 6-layer encoder's forward + backward alone in a hipGraph (8 token rows, d = 512, ffn 2048, dropout 0.1, parameters
in an arena as in the train step): what the 8-token section costs with nothing around it.
usage: python tools/probes/encoder_alone.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from vidsitu_amd import transformer_code as T
from vidsitu_amd.optim import ParamArena


def main():
    dev = torch.device("cuda", 0)
    for layers in (6, 3, 1):
        torch.manual_seed(0)
        mdl = T.Transformer(d_model=512, n_vocab_src=0, vocab_trg=0, d_hidden=2048, n_layers=layers, n_heads=8,
                            drop_ratio=0.1, pe=False).to(dev).train()
        arena = ParamArena(mdl)
        x = torch.randn(2, 4, 512, device=dev, requires_grad=True)
        dy = torch.randn(2, 4, 512, device=dev)

        def step():
            out = mdl.encoder(x)[-1]
            out.backward(dy)

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        best = 1e9
        for _ in range(10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / 5)
        # the same with 1.2 GB of streaming traffic in front of every replay (caches and TLBs cold, as behind the trunk)
        big = torch.empty(300_000_000, dtype=torch.float32, device=dev)
        cold = 1e9
        for _ in range(6):
            big.add_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            cold = min(cold, e0.elapsed_time(e1) * 1e3)
        del big
        print(f"{layers} layers: hot {best:7.1f} us per forward + backward ({best / (14 * layers):5.2f} us per launch at 14 per "
              f"layer); behind 2.4 GB of streaming traffic {cold:7.1f} us ({cold / (14 * layers):5.2f})")


if __name__ == "__main__":
    main()
