#!/bin/bash
# one-stream rocprofv3 kernel stats of the train step (eager, 5 steps) -> gpurun_out/r4_kstats/train_kernel_stats.csv
export TMPDIR=/tmp
OUT=gpurun_out/r4_kstats; mkdir -p $OUT
export VS_DUAL_STREAM=0 VS_WGRAD_LANES=0 VS_CONV_PAIR=0
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train -- python3 bench.py --steps 5 --warmup 2 --workload sf_txenc_train --no-cpu-baseline --no-roofline --no-feat-fwd --graph 0 > $OUT/rocprof_train.log 2>&1; echo "rocprof train exit $?"
f=$(find $OUT/prof_train -name "*kernel_stats*.csv" | head -1); cp "$f" $OUT/train_kernel_stats.csv; head -14 "$f" | cut -c1-160
find $OUT -name "*kernel_trace*.csv" -delete
