"""GPU frame resize (vs_resize_bicubic_u8) against PIL on the host cores: 8 clips x 32 frames of
360x640 RGB -> 224x224 (what the loader resizes per training step of 8 clips)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vidsitu_amd import ops

dev = torch.device("cuda:0")
x = torch.randint(0, 256, (8, 32, 360, 640, 3), dtype=torch.int32).to(torch.uint8)
xd = x.to(dev)
for _ in range(3):
    y = ops.resize_bicubic_u8(xd, 224, 224)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    y = ops.resize_bicubic_u8(xd, 224, 224)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"GPU: 256 frames 360x640 -> 224x224 in {ms:.3f} ms ({x.numel() / ms / 1e6:.1f} GB/s of source bytes)")
try:
    from PIL import Image
    fr = x.reshape(-1, 360, 640, 3).numpy()[:32]
    t0 = time.perf_counter()
    for f in fr:
        np.array(Image.fromarray(f).resize((224, 224)))
    dt = time.perf_counter() - t0
    print(f"PIL, 1 core: {dt / 32 * 1e3:.2f} ms per frame -> {dt / 32 * 256 * 1e3:.0f} ms per 256 frames")
except ImportError:
    pass
