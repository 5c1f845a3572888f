#!/bin/bash
# alternating A/B of one environment switch on the train step: tools/ab_env.sh VAR valueA valueB [reps] [pytest -k expr]
export TMPDIR=/tmp
OUT=gpurun_out/ab_env; mkdir -p $OUT
if [ -n "$5" ]; then
  timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -k "$5" > $OUT/pytest.log 2>&1; echo "pytest exit $?"; grep -E "AssertionError|Error|passed|failed" $OUT/pytest.log | cut -c1-300 | tail -5
fi
for i in $(seq 1 ${4:-3}); do
  for v in $2 $3; do
    env $1=$v timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1=$v', d['value'], d['ms_per_step'])" | tee -a $OUT/ab.log
  done
done
