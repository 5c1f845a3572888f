#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r2n; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_trunk.py tests/test_gpu_conv.py tests/test_gpu_dist_nccl.py -q -m gpu --no-header -p no:cacheprovider -x > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest.log
run() { name=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 5 > $OUT/$name.json 2> $OUT/$name.err; echo -n "$name: "; python -c "import json,sys; print(json.load(open('$OUT/$name.json'))['ms_per_step'])" 2>/dev/null || (echo fail; tail -3 $OUT/$name.err); }
run tail1 VS_WGRAD_TAIL=1
run tail0 VS_WGRAD_TAIL=0
run tail1b VS_WGRAD_TAIL=1
run tail0b VS_WGRAD_TAIL=0
run tail1_s512 VS_WGRAD_TAIL=1 VS_WGRAD_SLOTS=512
