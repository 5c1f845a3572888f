#!/bin/bash
# Round-4: deep weight gradients in the step: pair launches on / off x deep plan on / off, alternating.
export TMPDIR=/tmp
for rep in 1 2; do for pair in 1 0; do for a in 1 0; do
  VS_CONV_PAIR=$pair VS_WGRAD_DEEP=$a timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-feat-fwd 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train pair $pair wgrad_deep $a', d['value'], d['ms_per_step'])"
done; done; done
for a in 1 0; do
  VS_WGRAD_DEEP=$a timeout 600 python bench.py --clips-per-gpu 32 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-feat-fwd 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train 32 clips wgrad_deep $a', d['value'], d['ms_per_step'])"
done
