#!/bin/bash
# Round-4: (1) deep weight-gradient kernel with whole splits per XCD (VS_WGRAD_DEEP_ALIGN=1) vs not, 32 and 8 clips;
# (2) the GEMM probe's staggered all-wave copy issue against halves-burst.
export TMPDIR=/tmp
OUT=gpurun_out/r4_wgrad_align; mkdir -p $OUT
for c in 32 8; do for al in 1 0; do
  VS_WGRAD_DEEP_ALIGN=$al timeout 900 python tools/fwd_layer_times.py wgrad --clips=$c --only=s3.p0,s4.p0,s5.p0 --deep > $OUT/wgrad_${c}_align$al.txt 2>&1; echo "clips $c align $al: $(tail -1 $OUT/wgrad_${c}_align$al.txt)"
done; done
timeout 600 tools/probes/gemm_deep > $OUT/gemm_deep_v4.txt 2>&1; echo "probe exit $?"; cut -c1-250 $OUT/gemm_deep_v4.txt | grep -v check
