#!/bin/bash
# round-2 GPU session b: new tests, bench (train) health, cpu baseline timing, per-layer forward table, eval chunk A/B
export TMPDIR=/tmp
OUT=gpurun_out/r2b; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity_full.py tests/test_gpu_dist_nccl.py -q -m gpu -s --no-header -p no:cacheprovider > $OUT/pytest_new.log 2>&1; echo "pytest exit $?"; grep -E "passed|failed|rel_l2|logits|worst|^  [0-9]" $OUT/pytest_new.log | tail -40
timeout 400 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > $OUT/bench_train.json 2> $OUT/bench_train.err; echo "bench train exit $?"; head -c 700 $OUT/bench_train.json; echo; tail -3 $OUT/bench_train.err
( time timeout 500 python -c "
import bench, json
print(json.dumps(bench.cpu_baseline('sf_txenc_train', 1564)))
print(json.dumps(bench.cpu_baseline('feat_fwd', 1564)))
" ) > $OUT/cpu_baseline.log 2>&1; echo "cpu baseline exit $?"; tail -6 $OUT/cpu_baseline.log
for c in 1 2 4; do
  VS_EVAL_CHUNKS=$c timeout 300 python bench.py --workload feat_fwd --no-cpu-baseline --no-roofline --steps 30 --warmup 5 > $OUT/fwd_chunks$c.json 2> $OUT/fwd_chunks$c.err; echo "chunks $c exit $?"; head -c 260 $OUT/fwd_chunks$c.json; echo
done
timeout 600 python tools/fwd_layer_times.py fwd dgrad wgrad > $OUT/layer_times.txt 2>&1; echo "layers exit $?"; tail -60 $OUT/layer_times.txt
