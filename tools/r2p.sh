#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r2p; mkdir -p $OUT
run() { name=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 5 > $OUT/$name.json 2> $OUT/$name.err; echo -n "$name: "; python -c "import json,sys; print(json.load(open('$OUT/$name.json'))['ms_per_step'])" 2>/dev/null || (echo fail; tail -3 $OUT/$name.err); }
run base A=1
run two1024 VS_BN_TWO_LEVEL=1024
run two2048 VS_BN_TWO_LEVEL=2048
run two100000 VS_BN_TWO_LEVEL=100000
run base2 A=1
