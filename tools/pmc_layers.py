"""Launch the top conv GEMMs of the bench step (forward, dgrad, wgrad) REPS times each, in a fixed order,
and write the launch plan -- for the PMC passes of tools/pmc_mfma.sh, whose counter rows are matched to
layers by dispatch order.  usage: python tools/pmc_layers.py <plan.json> [reps]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidsitu_amd import ops

dev = torch.device("cuda:0")
plan_path = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
# name, Cin, T, H, W, Cout, k, s, p   (8 clips unless the name ends in "@N")
SHAPES = [
    ("s4.a 1024->256 [3,1,1]", 1024, 8, 14, 14, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    ("s4.b 256->256 [1,3,3]", 256, 8, 14, 14, 256, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("s4.c 256->1024 [1,1,1]", 256, 8, 14, 14, 1024, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("s5.a 2048->512 [3,1,1]", 2048, 8, 7, 7, 512, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    ("s5.b 512->512 [1,3,3]", 512, 8, 7, 7, 512, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("s3.b 128->128 [1,3,3]", 128, 8, 28, 28, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("s3.c 128->512 [1,1,1]", 128, 8, 28, 28, 512, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("s2.b 64->64 [1,3,3]", 64, 8, 56, 56, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("s2.c 64->256 [1,1,1]", 64, 8, 56, 56, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    # round 4: the launches the deep-pipeline kernels take (conv_deep.hip / conv_wgrad_deep_kernel)
    ("s4.b0.a 640->256 [3,1,1]", 640, 8, 28, 28, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    ("s4.a 1024->256 [3,1,1] @32", 1024, 8, 14, 14, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    ("s4.b 256->256 [1,3,3] @32", 256, 8, 14, 14, 256, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("s5.a 2048->512 [3,1,1] @32", 2048, 8, 7, 7, 512, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
]
plan = []
for name, cin, t, h, w, cout, k, s, p in SHAPES:
    nclips = int(name.split("@")[1]) if "@" in name else 8
    x = ops.new_act(nclips, cin, t, h, w, dev); x.normal_()
    wt = (torch.randn(cout, *k, cin, device=dev) / (cin * k[0] * k[1] * k[2]) ** 0.5).to(ops.BF16).permute(0, 4, 1, 2, 3)
    ys = ops.conv_out_shape(x.shape, cout, k, s, p)
    dy = ops.new_act(*ys, device=dev); dy.normal_()
    wtt = ops.weight_transpose(wt)
    dw = torch.empty((cout, *k, cin), dtype=torch.float32, device=dev).permute(0, 4, 1, 2, 3)
    fl = 2.0 * ys[0] * ys[2] * ys[3] * ys[4] * cout * cin * k[0] * k[1] * k[2]
    torch.cuda.synchronize()
    for kind in ("fwd", "dgrad", "wgrad"):
        for _ in range(reps):
            if kind == "fwd":
                ops.conv_fwd(x, wt, k, s, p, stats=True)
            elif kind == "dgrad":
                ops.conv_dgrad(dy, wtt, tuple(x.shape), k, s, p)
            else:
                ops.conv_wgrad(dy, x, k, s, p, out=dw)
        plan.append({"layer": name, "kind": kind, "launches": reps, "flops": fl})
    torch.cuda.synchronize()
json.dump(plan, open(plan_path, "w"))
print("plan written", len(plan))
