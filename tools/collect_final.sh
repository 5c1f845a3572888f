#!/bin/bash
# Copy the artefacts of `tools/ab.sh r6_final` (gpurun_out/r6_final/) into profiles/ under the round's names.
#   tools/collect_final.sh r06 v1
R=${1:-r06}; V=${2:-v1}; S=gpurun_out/r6_final; D=profiles
set -e
tail -n 1 $S/bench_train.json > $D/${R}_bench_train_${V}.json
tail -n 1 $S/bench_feat_fwd.json > $D/${R}_bench_feat_fwd_${V}.json
cp $S/train_kernel_stats_one_stream.csv $D/${R}_train_kernel_stats_${V}_one_stream.csv
cp $S/feat_fwd_kernel_stats_one_stream.csv $D/${R}_feat_fwd_kernel_stats_${V}_one_stream.csv
cp $S/pmc_traffic.json $D/pmc_traffic.json
cp $S/pmc_traffic_feat_fwd.json $D/pmc_traffic_feat_fwd.json
cp $S/pmc_mfma.json $D/pmc_mfma.json
tail -n 40 $S/pmc_mfma.log > $D/${R}_pmc_mfma_${V}.txt
cp $S/batch_probe.txt $D/${R}_batch_probe.txt
python tools/merge_parity.py "$(git rev-parse --short=12 HEAD)"
ls -la $D/${R}_* | head -20
