#!/bin/bash
# kernel-trace timeline statistics of the replayed step: tools/timeline.sh [workload]
export TMPDIR=/tmp; wl=${1:-sf_txenc_train}; mkdir -p gpurun_out/timeline
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$wl -- python3 bench.py --workload $wl --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-feat-fwd > gpurun_out/timeline/$wl.log 2>&1
f=$(find /tmp/tr_$wl -name "*kernel_trace.csv" | head -1)
python tools/trace_overlap.py "$f" | tee gpurun_out/timeline/${wl}_overlap.txt | tail -48
