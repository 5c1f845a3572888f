#!/bin/bash
# A/B of the layer-local parity test: previous build (tmp/prev) vs current
export TMPDIR=/tmp
OUT=gpurun_out/r3v; mkdir -p $OUT
VS_LIB_PATH=$PWD/tmp/prev/vidsitu_amd/libvidsitu_hip.so timeout 900 python -m pytest tests/test_gpu_parity_full.py -q -k every_resblock -s -p no:cacheprovider > $OUT/prev.log 2>&1; echo "prev exit $?"
timeout 900 python -m pytest tests/test_gpu_parity_full.py -q -k every_resblock -s -p no:cacheprovider > $OUT/cur.log 2>&1; echo "cur exit $?"
grep -A17 "per-block" $OUT/prev.log | tail -40
echo ======
grep -A17 "per-block" $OUT/cur.log | tail -40
