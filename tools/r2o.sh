#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r2o; mkdir -p $OUT
run() { name=$1; g=$2; shift; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 5 --graph $g > $OUT/$name.json 2> $OUT/$name.err; echo -n "$name: "; python -c "import json,sys; print(json.load(open('$OUT/$name.json'))['ms_per_step'])" 2>/dev/null || (echo fail; tail -3 $OUT/$name.err); }
run eager_tail0 0 VS_WGRAD_TAIL=0
run eager_tail1 0 VS_WGRAD_TAIL=1
run eager_lag1 0 VS_WGRAD_TAIL=0 VS_WGRAD_LAG=1
run eager_nolanes 0 VS_WGRAD_TAIL=0 VS_WGRAD_LANES=0
run eager_onestream 0 VS_WGRAD_TAIL=0 VS_WGRAD_LANES=0 VS_DUAL_STREAM=0
run graph_onestream 1 VS_WGRAD_TAIL=0 VS_WGRAD_LANES=0 VS_DUAL_STREAM=0
run graph_base 1 VS_WGRAD_TAIL=0
