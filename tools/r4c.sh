#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r4c; mkdir -p $OUT
for m in 3 0; do VS_TX_BAR=$m timeout 120 python tools/txstack_bar.py 2>&1 | grep -v amdgpu; done
timeout 900 python -m pytest tests/test_gpu_txenc.py -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest exit $?"; grep -E "AssertionError|passed|failed" $OUT/pytest.log | cut -c1-400
for i in 1 2 3; do VS_TX_BAR=3 timeout 600 python -m pytest tests/test_gpu_txenc.py -q -k stack -p no:cacheprovider 2>&1 | tail -1; done
