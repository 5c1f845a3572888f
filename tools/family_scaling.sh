#!/bin/bash
# Per-kernel-family time of the train step at two batch sizes (one stream, eager, rocprofv3 --kernel-trace --stats):
# which families carry the small-batch penalty?  usage: tools/family_scaling.sh [clipsA] [clipsB]  -> gpurun_out/family_scaling/
export TMPDIR=/tmp
OUT=gpurun_out/family_scaling; mkdir -p $OUT
export VS_DUAL_STREAM=0 VS_WGRAD_LANES=0
for c in ${1:-8} ${2:-32}; do
  rm -rf $OUT/prof_$c
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$c -- python3 bench.py --steps 5 --warmup 2 --clips-per-gpu $c --no-cpu-baseline --no-roofline --no-feat-fwd --graph 0 > $OUT/rocprof_$c.log 2>&1
  echo "rocprof $c clips exit $?"
  f=$(find $OUT/prof_$c -name "*kernel_stats*.csv" | head -1); cp "$f" $OUT/kernel_stats_$c.csv
  find $OUT/prof_$c -name "*kernel_trace*.csv" -delete
done
python3 tools/family_scaling.py $OUT/kernel_stats_${1:-8}.csv ${1:-8} $OUT/kernel_stats_${2:-32}.csv ${2:-32} | tee $OUT/table.txt
