#!/bin/bash
export TMPDIR=/tmp VS_DUAL_STREAM=0 VS_WGRAD_LANES=0 VS_CONV_PAIR=0
# one-stream rocprofv3 kernel stats of the training step -> gpurun_out/prof_train/kernel_stats.csv
OUT=gpurun_out/${1:-prof_train}
mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --steps 5 --warmup 2 --workload sf_txenc_train --no-cpu-baseline --no-roofline --graph 0 > $OUT/rocprof.log 2>&1
f=$(find $OUT/prof -name "*kernel_stats*.csv" | head -1); head -30 "$f" | cut -c1-160
cp "$f" $OUT/kernel_stats.csv
find $OUT/prof -name "*kernel_trace*.csv" -delete
