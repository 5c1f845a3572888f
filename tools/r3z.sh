#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r3z; mkdir -p $OUT
timeout 200 python tools/txstack_debug.py 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_gpu_txenc.py tests/test_gpu_gpt2.py tests/test_gpu_txdec.py -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest exit $?"; grep -E "AssertionError|passed|failed" $OUT/pytest.log | cut -c1-600
