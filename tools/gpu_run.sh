#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprof.  Everything lands in gpurun_out/.
set -u
OUT=gpurun_out/${1:-run}
mkdir -p $OUT
export TMPDIR=/tmp
echo "== rocm-smi" > $OUT/env.log; rocm-smi --showproductname 2>&1 | head -20 >> $OUT/env.log; nproc >> $OUT/env.log
echo "== pytest gpu"
timeout 1500 python -m pytest tests -q -m gpu --no-header -rA -p no:cacheprovider ${PYTEST_ARGS:-} > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log
tail -n 60 $OUT/pytest_gpu.log
echo "== smoke"
timeout 600 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/smoke.log; tail -n 5 $OUT/smoke.log
echo "== bench feat_fwd"
timeout 900 python bench.py --steps 10 --warmup 3 --workload feat_fwd > $OUT/bench_feat_fwd.log 2>&1; echo "bench exit $?" | tee -a $OUT/bench_feat_fwd.log; tail -n 4 $OUT/bench_feat_fwd.log
if [ "${RUN_TRAIN:-0}" = "1" ]; then
  echo "== bench train"
  timeout 900 python bench.py --steps 5 --warmup 2 --workload sf_txenc_train > $OUT/bench_train.log 2>&1; echo "bench exit $?" | tee -a $OUT/bench_train.log; tail -n 4 $OUT/bench_train.log
fi
if [ "${RUN_PROF:-1}" = "1" ]; then
  echo "== rocprof"
  # one stream: per-kernel durations of launches that do not overlap (what bench.py's roofline uses)
  export VS_DUAL_STREAM=0 VS_WGRAD_LANES=0
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --steps 5 --warmup 2 --workload ${PROF_WORKLOAD:-feat_fwd} --no-cpu-baseline --no-roofline --graph 0 > $OUT/rocprof.log 2>&1
  echo "rocprof exit $?" | tee -a $OUT/rocprof.log
  find $OUT/prof -name "*kernel_stats*.csv" | head -3
  f=$(find $OUT/prof -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && head -25 "$f"
  # keep the merge-back small: drop the per-dispatch trace, keep the stats
  find $OUT/prof -name "*kernel_trace*.csv" -size +20M -delete
fi
