export TMPDIR=/tmp
OUT=gpurun_out/r4_kstats32; mkdir -p $OUT
export VS_DUAL_STREAM=0 VS_WGRAD_LANES=0 VS_CONV_PAIR=0
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --steps 4 --warmup 2 --clips-per-gpu 32 --workload sf_txenc_train --no-cpu-baseline --no-roofline --no-feat-fwd --graph 0 > $OUT/log.txt 2>&1; echo exit $?
f=$(find $OUT/prof -name "*kernel_stats*.csv" | head -1); cp "$f" $OUT/train32_kernel_stats.csv; find $OUT -name "*kernel_trace*.csv" -delete
