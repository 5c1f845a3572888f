#!/bin/bash
# PMC counters for the conv microbenchmark (separate passes; no trace domains besides kernel-trace).
export TMPDIR=/tmp
OUT=gpurun_out/pmc_${1:-conv}
mkdir -p $OUT
python3 tools/conv_microbench.py 20 > $OUT/microbench.log 2>&1
cat $OUT/microbench.log
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/p1 -- python3 tools/conv_microbench.py 3 > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/p2 -- python3 tools/conv_microbench.py 3 > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/p3 -- python3 tools/conv_microbench.py 3 > $OUT/p3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/p4 -- python3 tools/conv_microbench.py 3 > $OUT/p4.log 2>&1
python3 - <<'PY'
import csv, glob, collections, os, sys
out = os.environ.get("OUT_DIR", "")
for pdir in sorted(glob.glob("gpurun_out/pmc_*/p[1-4]")):
    files = glob.glob(pdir + "/**/*counter_collection.csv", recursive=True)
    if not files: print(pdir, "no counter csv", glob.glob(pdir + "/**/*.csv", recursive=True)[:3]); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            if "conv_igemm" not in k: continue
            key = (k[:60], r.get("Grid_Size"))
            agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(key, r["Counter_Name"])] += 1
    print("==", pdir)
    for key, d in agg.items():
        print(key, {c: round(v / cnt[(key, c)], 1) for c, v in d.items()})
PY
find $OUT -name "*kernel_trace*.csv" -delete
