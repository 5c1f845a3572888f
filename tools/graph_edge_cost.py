"""What a cross-stream dependency costs inside a replayed hipGraph: N repetitions of
   A (main) -> fork -> [B (lane) || C (main)] -> join -> ...   against   A -> B -> C on one stream,
with kernels of a given duration (a spin kernel: torch.cuda._sleep cycles).  The difference per repetition is
the price of one fork + one join edge beyond same-stream boundaries (minus what running B beside C saves)."""
import os, sys
import torch

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
lane = torch.cuda.Stream()


def build(two_streams, cycles, n):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        torch.cuda._sleep(10)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            main = torch.cuda.current_stream()
            for _ in range(n):
                torch.cuda._sleep(cycles)  # A
                if two_streams:
                    lane.wait_stream(main)
                    with torch.cuda.stream(lane):
                        torch.cuda._sleep(cycles)  # B
                    torch.cuda._sleep(cycles)  # C
                    main.wait_stream(lane)
                else:
                    torch.cuda._sleep(cycles)
                    torch.cuda._sleep(cycles)
    return g


def time(g, reps=5):
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3)
    return best


n = 200
for cycles in (2000, 20000, 60000):  # ~1, ~10, ~30 us at ~2 GHz
    one = time(build(False, cycles, n)) / n
    two = time(build(True, cycles, n)) / n
    k = one / 3
    print(f"kernel ~{k:5.1f} us (incl. boundary): one stream {one:6.1f} us per repetition, fork/join {two:6.1f} us "
          f"-> ideal two-stream {2 * k:6.1f}, fork + join overhead {two - 2 * k:5.1f} us")
