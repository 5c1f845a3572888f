"""Aggregate the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_traffic.sh into bytes per launch
per kernel.  Units and corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB
of L2 memory-side requests; on gfx950 FETCH_SIZE reports exactly half the bytes of wide
(16 B/lane) coalesced reads, so it is doubled; WRITE_SIZE is exact for 16-B streaming stores."""
import collections
import csv
import glob
import json
import re
import sys

out_dir, workload = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
build = sys.argv[3] if len(sys.argv) > 3 else ""


def collect(sub, counter):
    acc, cnt = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(f"{out_dir}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
            acc[name] += float(r["Counter_Value"])
            cnt[name] += 1
    return acc, cnt


fetch, nf = collect("fetch", "FETCH_SIZE")
write, nw = collect("write", "WRITE_SIZE")
table = {}
for k in sorted(set(fetch) | set(write)):
    table[k] = {
        "launches": int(max(nf[k], nw[k])),
        "fetch_bytes": round(2.0 * 1024.0 * fetch[k] / max(nf[k], 1)),   # x2: gfx950 correction
        "write_bytes": round(1024.0 * write[k] / max(nw[k], 1)),
    }
meta = {"_meta": {"workload": workload, "build": build, "unit": "bytes per launch (average over the launches of one step)",
                  "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes); "
                            "FETCH_SIZE x 1024 x 2, WRITE_SIZE x 1024"}}
meta.update(table)
json.dump(meta, open(f"{out_dir}/pmc_traffic.json", "w"), indent=1, sort_keys=True)
tot = sorted(table.items(), key=lambda kv: -(kv[1]["fetch_bytes"] + kv[1]["write_bytes"]) * kv[1]["launches"])
for k, v in tot[:25]:
    print(f"{k[:70]:70s} x{v['launches']:4d}  fetch {v['fetch_bytes']/1e6:9.2f} MB  write {v['write_bytes']/1e6:9.2f} MB")
