#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r2i; mkdir -p $OUT
run() { name=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 5 > $OUT/$name.json 2> $OUT/$name.err; echo -n "$name: "; python -c "import json,sys; print(json.load(open('$OUT/$name.json'))['ms_per_step'])" 2>/dev/null || echo fail; }
run base A=1
run slots192 VS_WGRAD_SLOTS=192
run slots256 VS_WGRAD_SLOTS=256
run slots320 VS_WGRAD_SLOTS=320
run slots512 VS_WGRAD_SLOTS=512
run nolanes VS_WGRAD_LANES=0
run nofuse VS_FUSE_BN_SUMS=0
run base2 A=1
run halo2 VS_CONV_HALO=2
run halo0 VS_CONV_HALO=0
