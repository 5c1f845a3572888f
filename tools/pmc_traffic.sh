#!/bin/bash
# HBM traffic of every kernel of the bench step from PMC counters, as MI355X_MICROARCH.md
# prescribes: separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass), no trace
# domain besides --kernel-trace, the program itself after `--`.
#   bash tools/pmc_traffic.sh [workload]   -> gpurun_out/pmc_traffic/{pmc_traffic.json,*.log}
export TMPDIR=/tmp
export VS_DUAL_STREAM=0 VS_WGRAD_LANES=0 VS_CONV_PAIR=0
WL=${1:-sf_txenc_train}
OUT=gpurun_out/pmc_traffic
mkdir -p $OUT
ARGS="bench.py --steps 1 --warmup 1 --workload $WL --graph 0 --no-cpu-baseline --no-roofline"
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ARGS > $OUT/fetch.log 2>&1
echo "fetch pass exit $?"
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ARGS > $OUT/write.log 2>&1
echo "write pass exit $?"
python3 tools/pmc_traffic.py $OUT $WL "${VS_BUILD_TAG:-untagged}"
find $OUT -name "*kernel_trace*.csv" -delete
find $OUT -name "*counter_collection*.csv" -size +30M -delete
