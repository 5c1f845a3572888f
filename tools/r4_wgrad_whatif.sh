#!/bin/bash
# Round-4: (1) the train step WITHOUT its weight gradients (VS_WHATIF=4, garbage dW: timing only) against the real step;
# (2) all weight gradients of a step packed back to back on 1-4 streams with nothing in their way.
export TMPDIR=/tmp
for rep in 1 2; do for w in 0 4; do
  VS_WHATIF=$w timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-feat-fwd 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train whatif $w', d['value'], d['ms_per_step'])"
done; done
timeout 600 python tools/wgrad_packed.py 1,2,3,4
