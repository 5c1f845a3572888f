#!/bin/bash
# Round-4 batch probe: the same step at 8 / 16 / 32 / 40 / 64 clips per GPU (8 = the BASELINE config; 40 = the reference's
# default train.bs x 5 events; the others are probes), deep kernels on / off.
export TMPDIR=/tmp
for n in 8 16 32 40 64; do for deep in 1 0; do
  for wl in sf_txenc_train feat_fwd; do
    VS_CONV_DEEP=$deep VS_WGRAD_DEEP=$deep timeout 600 python bench.py --workload $wl --clips-per-gpu $n --steps 15 --warmup 4 --no-cpu-baseline --no-roofline --no-feat-fwd 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl clips $n deep $deep', d['value'], d['ms_per_step'], d['config'].get('frac_of_bf16_mfma_peak'))"
  done
done; done
