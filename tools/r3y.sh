#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out/r3y
for wl in sf_txenc_train feat_fwd; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$wl -- python3 bench.py --workload $wl --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/r3y/$wl.log 2>&1
  f=$(find /tmp/tr_$wl -name "*kernel_trace.csv" | head -1)
  echo "== $wl"; python tools/trace_overlap.py "$f" | tee gpurun_out/r3y/${wl}_overlap.txt | tail -40
done
