#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r3_c; mkdir -p $OUT
timeout 2400 python -m pytest tests/test_gpu_dist_two_ranks.py tests/test_gpu_parity_full.py -q -m gpu --no-header -p no:cacheprovider -s > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest.log
grep -E "relative|decomposition|split bf16|passed|failed|FAILED|Error|differ" $OUT/pytest.log | cut -c1-900 | head -40
for v in 0 1; do
  env VS_EVAL_SPLIT_WEIGHTS=$v timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 100 --workload feat_fwd 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fwd VS_EVAL_SPLIT_WEIGHTS=$v', d['value'], d['ms_per_step'])" | tee -a $OUT/ab.log
done
# weight gradients: unsplit where the output alone has >= N tiles; resident-slot target
for cfg in "A=0" "VS_WGRAD_S1_TILES=128" "A=0" "VS_WGRAD_S1_TILES=128" "VS_WGRAD_S1_TILES=96" "VS_WGRAD_SLOTS=256" "VS_WGRAD_SLOTS=256 VS_WGRAD_S1_TILES=128" "VS_WGRAD_SLOTS=192 VS_WGRAD_S1_TILES=128" "A=0"; do
  env $cfg timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['value'], d['ms_per_step'])" | tee -a $OUT/wgrad_s1.log
done
