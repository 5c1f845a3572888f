#!/bin/bash
# Round-4 what-if table: the replayed train step with whole kernel families skipped (VS_WHATIF bits: 1 bn_finalize, 2 bn_bwd_finalize,
# 4 conv_wgrad + slab reduce, 8 bn_apply / bn_bwd_reduce / bn_bwd_apply).  Garbage numerics: timing only.
export TMPDIR=/tmp
for rep in 1 2; do for w in 0 4 8 11 15; do
  VS_WHATIF=$w timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-feat-fwd 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train whatif $w', d['value'], d['ms_per_step'])"
done; done
