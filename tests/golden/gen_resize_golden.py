"""Golden vectors of the frame resize (run once in the build container; needs Pillow):
outputs of `PIL.Image.fromarray(x).resize((w, h))` -- the call of `VsituDS.read_img`
(`vidsitu_code/dat_loader.py:183-191`) -- on seeded random and smooth RGB images."""
import os

import numpy as np
import PIL
from PIL import Image

rs = np.random.RandomState(7)
cases = {}
for name, (h, w, oh, ow) in {"down": (45, 80, 28, 28), "down_odd": (61, 37, 24, 20), "up": (10, 12, 24, 24),
                             "same_w": (50, 24, 24, 24), "frame": (90, 160, 56, 56)}.items():
    x = rs.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
    if name == "frame":  # smooth content with saturated patches (exercises the clip after each pass)
        yy, xx = np.mgrid[0:h, 0:w]
        x = np.stack([(np.sin(xx / 7.0) * 127 + 128), (np.cos(yy / 5.0) * 127 + 128), (xx + yy) % 256], -1)
        x = x.clip(0, 255).astype(np.uint8)
        x[20:40, 30:70] = 255
        x[50:60, 100:150] = 0
    y = np.array(Image.fromarray(x).resize((ow, oh)))
    cases[name + "_in"], cases[name + "_out"] = x, y
cases["pillow_version"] = np.array(PIL.__version__)
np.savez_compressed(os.path.join(os.path.dirname(__file__), "resize_u8.npz"), **cases)
print("wrote resize_u8.npz with Pillow", PIL.__version__)
