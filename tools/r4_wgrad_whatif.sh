#!/bin/bash
# Round-4: (1) the train step WITHOUT its weight gradients (VS_WHATIF=4, garbage dW: timing only) against the real step;
# (a packed run of all weight gradients on 1-4 streams in one hipGraph was tried and died in hipStreamEndCapture: profiles/r04_wgrad_whatif.txt)
export TMPDIR=/tmp
for rep in 1 2; do for w in 0 4; do
  VS_WHATIF=$w timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-feat-fwd 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train whatif $w', d['value'], d['ms_per_step'])"
done; done

