"""Timeline statistics of one replayed step from a rocprofv3 --kernel-trace CSV: wall span, summed kernel time, busy
time (>= 1 kernel running), overlapped time (>= 2), idle gaps, and the kernels that follow the longest gaps.
usage: python tools/trace_overlap.py kernel_trace.csv [n_last_kernels_per_step]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda x: x[0])
# steps are separated by the largest idle gaps (host sync between replays): split at gaps > 200 us
steps, cur = [], [ks[0]]
for a, b in zip(ks, ks[1:]):
    if b[0] - max(x[1] for x in cur[-8:]) > 200_000:
        steps.append(cur); cur = []
    cur.append(b)
steps.append(cur)
big = [s for s in steps if len(s) > 50]
print(f"{len(ks)} dispatches, {len(steps)} bursts, {len(big)} with > 50 kernels")
for s in big[-3:]:
    t0, t1 = s[0][0], max(x[1] for x in s)
    ev = []
    for a, b, _ in s:
        ev.append((a, 1)); ev.append((b, -1))
    ev.sort()
    busy = over = 0; depth = 0; last = t0
    for t, d in ev:
        if depth >= 1: busy += t - last
        if depth >= 2: over += t - last
        depth += d; last = t
    tot = sum(b - a for a, b, _ in s)
    print(f"step: {len(s)} kernels, span {(t1 - t0) / 1e3:.1f} us, sum {tot / 1e3:.1f} us, busy {busy / 1e3:.1f} us, "
          f"overlapped {over / 1e3:.1f} us, idle {(t1 - t0 - busy) / 1e3:.1f} us")
s = big[-1]
agg = collections.defaultdict(lambda: [0, 0])
for a, b, n in s:
    k = n.split("(")[0][:70]
    agg[k][0] += b - a; agg[k][1] += 1
print("top kernels by in-situ time (last step):")
for k, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"  {t / 1e3:8.1f} us {c:4d} x {t / c / 1e3:7.1f}  {k}")

# the encoder / head section of a train step: from the end of the last average pool (forward) to the start of the
# first average-pool backward -- one stream, nothing else running
names = [n for _, _, n in s]
try:
    i1 = min(i for i, n in enumerate(names) if n.startswith("avgpool_bwd"))
    i0 = max(i for i, n in enumerate(names[:i1]) if n.startswith("avgpool_fwd"))
    sec = s[i0 + 1:i1]
    span = s[i1][0] - s[i0][1]
    busy = sum(b - a for a, b, _ in sec)
    print(f"encoder / head section: {len(sec)} kernels, span {span / 1e3:.1f} us, kernel time {busy / 1e3:.1f} us, "
          f"gaps {(span - busy) / 1e3:.1f} us")
    agg2 = collections.defaultdict(lambda: [0, 0])
    for a, b, n in sec:
        k = n.split("(")[0][:60]
        agg2[k][0] += b - a; agg2[k][1] += 1
    for k, (t, c) in sorted(agg2.items(), key=lambda kv: -kv[1][0])[:14]:
        print(f"  {t / 1e3:7.1f} us {c:3d} x {t / c / 1e3:6.1f}  {k}")
except ValueError:
    pass

# idle time (no kernel running) attributed to the kernel that ends the gap, and to the one that ran before it
ev = sorted(s, key=lambda x: x[0])
end_max, prev_name = ev[0][1], ev[0][2]
by_next, by_pair = collections.defaultdict(lambda: [0, 0]), collections.defaultdict(lambda: [0, 0])
for a, b, n in ev[1:]:
    if a > end_max:
        k = n.split("(")[0][:50]
        by_next[k][0] += a - end_max; by_next[k][1] += 1
        kp = prev_name.split("(")[0][:40] + " -> " + k[:40]
        by_pair[kp][0] += a - end_max; by_pair[kp][1] += 1
    if b > end_max:
        end_max, prev_name = b, n
print("idle before (last burst):")
for k, (t, c) in sorted(by_next.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"  {t / 1e3:7.1f} us {c:4d} gaps {t / c / 1e3:5.1f} us each  {k}")
print("idle by (kernel that ended last -> kernel that starts):")
for k, (t, c) in sorted(by_pair.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"  {t / 1e3:7.1f} us {c:4d} x {t / c / 1e3:5.1f}  {k}")

# how many kernels run at once (time at depth 0 / 1 / 2 / 3+), over the last step, and which kernels run ALONE
ev2 = []
for a, b, n in s:
    ev2.append((a, 1, n)); ev2.append((b, -1, n))
ev2.sort(key=lambda x: (x[0], x[1]))
hist = collections.Counter(); alone = collections.defaultdict(int); live = {}
last = ev2[0][0]; depth = 0
for t, d, n in ev2:
    dt = t - last
    if dt > 0:
        hist[min(depth, 3)] += dt
        if depth == 1:
            alone[next(iter(live.values())).split("(")[0][:60]] += dt
    if d > 0:
        live[(t, n)] = n
    else:
        for k in list(live):
            if k[1] == n:
                del live[k]; break
    depth += d; last = t
span = (max(x[1] for x in s) - s[0][0])
print("time by number of kernels running (last step): " + ", ".join(f"{k}{'+' if k == 3 else ''}: {v / 1e3:.0f} us ({100.0 * v / span:.0f} %)" for k, v in sorted(hist.items())))
print("kernels that run ALONE on the chip (no other kernel resident), by time:")
for k, t in sorted(alone.items(), key=lambda kv: -kv[1])[:16]:
    print(f"  {t / 1e3:8.1f} us  {k}")

# torch-native kernels inside the replayed step (fills, copies, elementwise glue): are the PMC pass's FillFunctor /
# transposes part of the steady-state step?
nat = collections.defaultdict(lambda: [0, 0])
for a, b, n in s:
    if "at::native" in n or "transpose" in n or "cast_f32" in n:
        k = n.replace("void ", "")[:110]
        nat[k][0] += b - a; nat[k][1] += 1
print("torch-native / layout kernels in the last burst (time, launches):")
for k, (t, c) in sorted(nat.items(), key=lambda kv: -kv[1][0])[:12]:
    print(f"  {t / 1e3:8.1f} us {c:4d} x  {k}")
