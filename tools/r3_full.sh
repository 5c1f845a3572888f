#!/bin/bash
# Round-3 artefact run: full GPU test suite, smoke, both bench workloads (default flags), the driver's launch forms.
export TMPDIR=/tmp
TAG=${1:-r03}
OUT=gpurun_out/r3_full; mkdir -p $OUT
export VS_BUILD_TAG="$TAG"
timeout 3000 python -m pytest tests -q -m gpu --no-header -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log; tail -6 $OUT/pytest_gpu.log | cut -c1-300
timeout 600 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -3 $OUT/smoke.log
( time timeout 900 python bench.py ) > $OUT/bench_train.json 2> $OUT/bench_train.err; echo "bench train exit $?"; head -c 400 $OUT/bench_train.json; echo; tail -4 $OUT/bench_train.err
( time timeout 600 python bench.py --workload feat_fwd ) > $OUT/bench_feat_fwd.json 2> $OUT/bench_feat_fwd.err; echo "bench fwd exit $?"; head -c 300 $OUT/bench_feat_fwd.json; echo
# the driver's N > 1 launch form, with one rank (what a 1-GPU box can run)
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/bench_torchrun1.json 2> $OUT/bench_torchrun1.err; echo "torchrun form exit $?"; head -c 300 $OUT/bench_torchrun1.json; echo
timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 > $OUT/bench_gpus2.out 2> $OUT/bench_gpus2.err; echo "bench --gpus 2 on this box: exit $? (non-zero expected)"; tail -2 $OUT/bench_gpus2.err
