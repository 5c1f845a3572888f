#!/bin/bash
# A/B of two builds of the library in one GPU session: tmp/prev (git archive of a commit, built with make) vs the tree
export TMPDIR=/tmp
OUT=gpurun_out/ab_lib; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_txenc.py tests/test_gpu_gpt2.py tests/test_gpu_txdec.py tests/test_gpu_train_step.py tests/test_gpu_dist_nccl.py -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest exit $?"; grep -E "AssertionError|passed|failed" $OUT/pytest.log | cut -c1-400
for i in 1 2 3; do
  for which in prev cur; do
    if [ $which = prev ]; then export VS_LIB_PATH=$PWD/tmp/prev/vidsitu_amd/libvidsitu_hip.so; else unset VS_LIB_PATH; fi
    timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which', d['value'], d['ms_per_step'])" | tee -a $OUT/ab.log
  done
done
