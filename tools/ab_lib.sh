#!/bin/bash
# alternating A/B of two builds of the library on the train step and the eval forward, one gpurun call:
#   tools/ab_lib.sh OLD.so NEW.so [reps] ["pytest -k expr" run on the NEW (default) library first]
export TMPDIR=/tmp
OUT=gpurun_out/ab_lib; mkdir -p $OUT
if [ -n "$4" ]; then
  timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider -k "$4" > $OUT/pytest.log 2>&1; echo "pytest exit $?"; grep -E "AssertionError|Error|passed|failed" $OUT/pytest.log | cut -c1-300 | tail -8
fi
for i in $(seq 1 ${3:-3}); do
  for lib in "$1" "$2"; do
    env VS_LIB_PATH=$lib timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-feat-fwd --steps 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train lib=$lib', d['value'], d['ms_per_step'])" | tee -a $OUT/ab.log
  done
done
for i in $(seq 1 2); do
  for lib in "$1" "$2"; do
    env VS_LIB_PATH=$lib timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 200 --workload feat_fwd 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fwd lib=$lib', d['value'], d['ms_per_step'])" | tee -a $OUT/ab.log
  done
done
