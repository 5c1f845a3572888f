"""hipGraph capture and nested forks (round 6): a stream forked from the capturing stream that itself forks a child
(or exchanges an event with another forked stream) crashes the process at capture end on this HIP build;
forks from the origin stream alone are fine.  usage: python tools/probes/graph_nested_fork.py noside|nested|pre LEVELS"""
import torch, sys
mode = sys.argv[1]
dev = torch.device("cuda:0")
x = torch.ones(1 << 20, device=dev)
B, sA, sB = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
def work(t, k):
    for _ in range(k): t = t * 1.0001
    return t
def fwd(levels):
    main = torch.cuda.current_stream()
    B.wait_stream(main)
    if mode == "pre":
        sB.wait_stream(main); sA.wait_stream(main)
    outs = []
    for rep in range(levels):
        sA.wait_stream(main)
        with torch.cuda.stream(sA): a = work(x, 3)
        b = work(x, 3)
        main.wait_stream(sA)
        outs += [a, b]
    with torch.cuda.stream(B):
        for rep in range(levels):
            if mode != "noside":
                sB.wait_stream(B)
                with torch.cuda.stream(sB): c = work(x, 3)
            else:
                c = work(x, 3)
            d = work(x, 3)
            if mode != "noside":
                B.wait_stream(sB)
            outs += [c, d]
    main.wait_stream(B)
    if mode == "pre":
        main.wait_stream(sB); main.wait_stream(sA)
    return outs
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side): fwd(int(sys.argv[2]))
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    o = fwd(int(sys.argv[2]))
g.replay(); torch.cuda.synchronize()
print(mode, sys.argv[2], "capture ok", float(o[0][0]))
