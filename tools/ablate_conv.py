import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidsitu_amd import ops
dev = torch.device("cuda:0")
SH = [("s4.a 1024->256 [3,1,1]", 8, 1024, 8, 14, 14, 256, (3,1,1), (1,1,1), (1,0,0)),
      ("s4.b 256->256 [1,3,3]", 8, 256, 8, 14, 14, 256, (1,3,3), (1,1,1), (0,1,1)),
      ("s5.a 2048->512 [3,1,1]", 8, 2048, 8, 7, 7, 512, (3,1,1), (1,1,1), (1,0,0)),
      ("big  512->512 [1,3,3] M=100352", 8, 512, 8, 56, 28, 512, (1,3,3), (1,1,1), (0,1,1))]
for name, n, cin, t, h, w, cout, k, s, p in SH:
    x = ops.new_act(n, cin, t, h, w, dev); x.normal_()
    wt = (torch.randn(cout, *k, cin, device=dev) / (cin*k[0]*k[1]*k[2])**0.5).to(ops.BF16).permute(0,4,1,2,3)
    ys = ops.conv_out_shape(x.shape, cout, k, s, p)
    flops = 2.0*ys[0]*ys[2]*ys[3]*ys[4]*cout*cin*k[0]*k[1]*k[2]
    row = f"{name:34s}"
    for tile, dbg, lab in [(None,0,"auto"), (0,0,"128x128"), (0,1,"no-gload"), (0,2,"no-mfma"), (0,3,"no-sstore"), (0,4,"no-ldsread")]:
        fn = lambda: ops.conv_fwd(x, wt, k, s, p, tile=tile, dbg=dbg)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)/20
        row += f" | {lab}: {ms*1e3:6.1f}us {flops/ms/1e9:6.0f}TF"
    print(row)
