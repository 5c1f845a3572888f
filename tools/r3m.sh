#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
for rep in 1 2; do for cfg in "X=1" "VS_CONV_PW=0" "VS_CONV_PW=2"; do
  env $cfg timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2> /dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/$cfg rep$rep /"
done; done
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python tools/trace_overlap.py $(find /tmp/tr -name "*kernel_trace.csv" | head -1) | tail -18
