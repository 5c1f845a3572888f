#!/bin/bash
# Round-4: full GPU suite; halo planner's conflict price A/B (padded lines vs conflicting fragment reads);
# matrix-core busy / stall / LDS-conflict counters incl. the launches of the deep-pipeline kernels.
export TMPDIR=/tmp
OUT=gpurun_out/r4_halo_pmc; mkdir -p $OUT
timeout 3000 python -m pytest tests -q -m gpu --no-header -p no:cacheprovider -x > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest_gpu.log
for wgt in 0.85 0.6; do
  VS_HALO_CONFLICT_WEIGHT=$wgt timeout 600 python tools/fwd_layer_times.py fwd dgrad --only=s4.p0.b1.b,s5.p0.b1.b,s4.p0.b0.b,s5.p0.b0.b,s3.p0.b1.b > $OUT/halo_w$wgt.txt 2>&1; echo "halo weight $wgt:"; grep "^s" $OUT/halo_w$wgt.txt | cut -c1-110
done
VS_BUILD_TAG=r04-v1 bash tools/pmc_mfma.sh > $OUT/pmc_mfma.log 2>&1; tail -45 $OUT/pmc_mfma.log; cp gpurun_out/pmc_mfma/pmc_mfma.json $OUT/pmc_mfma.json
