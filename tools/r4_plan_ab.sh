#!/bin/bash
# conv tests + trunk tests on the new plan rules, then the step A/B against the previous build
export TMPDIR=/tmp
OUT=gpurun_out/r4_plan; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_conv.py tests/test_gpu_trunk.py -q -m gpu --no-header -p no:cacheprovider -x > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest.log; tail -4 $OUT/pytest.log
bash tools/r4_ab_lib.sh $1 8
