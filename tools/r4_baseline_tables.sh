#!/bin/bash
# Round-4 first GPU session: the fill-path ceiling (LDS-DMA rate per CU by serving level, waves and depth) and the
# per-layer tables of the round-3 build at 8 clips (the bench shape) and 32 clips (each kernel's steady state).
export TMPDIR=/tmp
OUT=gpurun_out/r4_baseline; mkdir -p $OUT
timeout 300 tools/probes/lds_dma_rate > $OUT/lds_dma_rate.txt 2>&1; echo "dma probe exit $?"; head -30 $OUT/lds_dma_rate.txt
timeout 900 python tools/fwd_layer_times.py fwd dgrad wgrad > $OUT/layer_times_8.txt 2>&1; echo "layers8 exit $?"; tail -4 $OUT/layer_times_8.txt
timeout 900 python tools/fwd_layer_times.py fwd dgrad wgrad --clips=32 --only=s2.p0,s3.p0,s4.p0,s5.p0 > $OUT/layer_times_32.txt 2>&1; echo "layers32 exit $?"; tail -4 $OUT/layer_times_32.txt
