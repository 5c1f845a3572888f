#!/bin/bash
# PMC traffic table of the forward workload + both bench lines (checks the kernel labels against the PMC names)
export TMPDIR=/tmp VS_BUILD_TAG=r02-v8
bash tools/pmc_traffic.sh feat_fwd > gpurun_out/pmc_fwd.log 2>&1; tail -3 gpurun_out/pmc_fwd.log
cp gpurun_out/pmc_traffic/pmc_traffic.json gpurun_out/pmc_traffic_feat_fwd.json
cp gpurun_out/pmc_traffic_feat_fwd.json profiles/pmc_traffic_feat_fwd.json
timeout 600 python bench.py --workload feat_fwd --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_fwd_line.json; python -c "
import json; d=json.load(open('gpurun_out/bench_fwd_line.json')); r=d['roofline']; print(d['value'], r['kernel'], r['frac'], r['traffic'], r['algorithmic_bytes_per_launch'])"
timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_train_line.json; python -c "
import json; d=json.load(open('gpurun_out/bench_train_line.json')); r=d['roofline']; print(d['value'], r['kernel'][:40], r['frac'], r['traffic'], r['algorithmic_bytes_per_launch']); print(json.dumps(r.get('families', r.get('all_conv')))[:600])"
