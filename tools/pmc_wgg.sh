#!/bin/bash
# fabric-side traffic and L2 hit rate of the grouped weight-gradient launch under both block mappings
# (VS_WGG_JOBS=0: round-5 problem ranges, 1: jobs dealt to XCDs in full 32-block rounds): separate --pmc passes.
export TMPDIR=/tmp
OUT=gpurun_out/pmc_wgg; mkdir -p $OUT
for v in 0 1; do
  for c in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $c | tr ' ' '_')
    VS_WGG_JOBS=$v timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/j${v}_$tag -- python3 tools/wgrad_group_time.py --span=3 > $OUT/j${v}_$tag.log 2>&1
  done
done
python3 - <<'PY'
import csv, glob, collections
for v in (0, 1):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(f"gpurun_out/pmc_wgg/j{v}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "wgrad" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
            if r["Counter_Name"] == "FETCH_SIZE": n[k] += 1
    for k, d in acc.items():
        hit, miss = d.get("TCC_HIT_sum", 0), d.get("TCC_MISS_sum", 0)
        print(f"VS_WGG_JOBS={v} {k[:50]:50s} launches {n[k]:4d} fetch/launch {2*1024*d.get('FETCH_SIZE',0)/max(n[k],1)/1e6:8.1f} MB  L2 hit rate {hit/max(hit+miss,1):.3f}")
PY
find $OUT -name "*kernel_trace*.csv" -delete; find $OUT -name "*counter_collection*.csv" -size +10M -delete
