#!/bin/bash
# Round-4 artefact run: full GPU test suite, smoke, both bench workloads (default flags), one-stream rocprofv3
# kernel stats, PMC traffic and MFMA-busy passes.  Everything lands in gpurun_out/r4_final/.
export TMPDIR=/tmp
TAG=${1:-r04}
OUT=gpurun_out/r4_final; mkdir -p $OUT
export VS_BUILD_TAG="$TAG"
timeout 2400 python -m pytest tests -q -m gpu --no-header -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
timeout 600 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -3 $OUT/smoke.log
( time timeout 900 python bench.py ) > $OUT/bench_train.json 2> $OUT/bench_train.err; echo "bench train exit $?"; head -c 300 $OUT/bench_train.json; echo; tail -4 $OUT/bench_train.err
( time timeout 600 python bench.py --workload feat_fwd ) > $OUT/bench_feat_fwd.json 2> $OUT/bench_feat_fwd.err; echo "bench fwd exit $?"; head -c 300 $OUT/bench_feat_fwd.json; echo
export VS_DUAL_STREAM=0 VS_WGRAD_LANES=0 VS_CONV_PAIR=0
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train -- python3 bench.py --steps 5 --warmup 2 --workload sf_txenc_train --no-cpu-baseline --no-roofline --graph 0 > $OUT/rocprof_train.log 2>&1; echo "rocprof train exit $?"
f=$(find $OUT/prof_train -name "*kernel_stats*.csv" | head -1); cp "$f" $OUT/train_kernel_stats_one_stream.csv; head -12 "$f" | cut -c1-150
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_fwd -- python3 bench.py --steps 5 --warmup 2 --workload feat_fwd --no-cpu-baseline --no-roofline --graph 0 > $OUT/rocprof_fwd.log 2>&1; echo "rocprof fwd exit $?"
f=$(find $OUT/prof_fwd -name "*kernel_stats*.csv" | head -1); cp "$f" $OUT/feat_fwd_kernel_stats_one_stream.csv
find $OUT -name "*kernel_trace*.csv" -delete
unset VS_DUAL_STREAM VS_WGRAD_LANES VS_CONV_PAIR
bash tools/pmc_traffic.sh sf_txenc_train > $OUT/pmc_traffic.log 2>&1; tail -12 $OUT/pmc_traffic.log; cp gpurun_out/pmc_traffic/pmc_traffic.json $OUT/pmc_traffic.json
bash tools/pmc_mfma.sh > $OUT/pmc_mfma.log 2>&1; tail -28 $OUT/pmc_mfma.log; cp gpurun_out/pmc_mfma/pmc_mfma.json $OUT/pmc_mfma.json
