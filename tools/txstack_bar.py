"""Barrier cost of vs_txenc_stack_run: N trivial stages (a 32 KB add) in one launch, graph replay."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidsitu_amd import ops
dev = torch.device("cuda", 0)
a, b, y = (torch.randn(8, 1024, device=dev) for _ in range(3))
for n in (1, 33, 65):
    st = ops.TxStack(dev)
    for _ in range(n):
        st.add(a, b, y)
    st.run(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10):
            st.run()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print(f"VS_TX_BAR={os.environ.get('VS_TX_BAR','0')} stages {n:3d}: {e0.elapsed_time(e1) * 100:.1f} us per launch, failed={st.failed()}")
