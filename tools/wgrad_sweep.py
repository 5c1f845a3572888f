"""Sweep of the weight-gradient plan (tile x block slots) on the slow s3-s5 shapes of the bench step; each launch
(+ its slab reduce) alone on the GPU, replayed from a hipGraph.  usage: python tools/wgrad_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidsitu_amd import ops
dev = torch.device("cuda:0")
REPS = 20

def gt(fn):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REPS): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / REPS)
    return best * 1e3

SH = [
    ("s4.a 1024->256 [3,1,1]", 1024, 8, 14, 14, 256, (3, 1, 1), (1, 0, 0)),
    ("s4.b 256->256 [1,3,3]", 256, 8, 14, 14, 256, (1, 3, 3), (0, 1, 1)),
    ("s4.c 256->1024 [1,1,1]", 256, 8, 14, 14, 1024, (1, 1, 1), (0, 0, 0)),
    ("s5.a 2048->512 [3,1,1]", 2048, 8, 7, 7, 512, (3, 1, 1), (1, 0, 0)),
    ("s5.b 512->512 [1,3,3]", 512, 8, 7, 7, 512, (1, 3, 3), (0, 1, 1)),
    ("s5.c 512->2048 [1,1,1]", 512, 8, 7, 7, 2048, (1, 1, 1), (0, 0, 0)),
    ("s3.b 128->128 [1,3,3]", 128, 8, 28, 28, 128, (1, 3, 3), (0, 1, 1)),
    ("s3.c 128->512 [1,1,1]", 128, 8, 28, 28, 512, (1, 1, 1), (0, 0, 0)),
]
def main():
  for name, cin, t, h, w, cout, k, p in SH:
      x = ops.new_act(8, cin, t, h, w, dev); x.normal_()
      dy = ops.new_act(8, cout, t, h, w, dev); dy.normal_()
      dw = torch.empty((cout, *k, cin), dtype=torch.float32, device=dev).permute(0, 4, 1, 2, 3)
      row = f"{name:24s} plan {gt(lambda: ops.conv_wgrad(dy, x, k, (1, 1, 1), p, out=dw)):6.1f} |"
      for tile in (0, 1, 2, 3):
          for slots in (192, 256, 384, 512):
              try:
                  us = gt(lambda: ops.conv_wgrad(dy, x, k, (1, 1, 1), p, out=dw, tile=tile, slots=slots))
                  row += f" t{tile}s{slots}:{us:6.1f}"
              except Exception as e:
                  row += f" t{tile}s{slots}:  ERR"
          row += " |"
      print(row, flush=True)


if __name__ == "__main__":
    main()
