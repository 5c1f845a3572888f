#!/bin/bash
# Round-4: (1) chunk-major vs tap-major reduction order per layer (VS_CONV_KORDER_MIN=2 / 0), 8 and 32 clips;
# (2) the replayed train step's timeline (kernels running at once, who runs alone, idle attribution); (3) step A/B.
export TMPDIR=/tmp
OUT=gpurun_out/r4_korder; mkdir -p $OUT
for c in 8 32; do
  for k in 2 0; do
    VS_CONV_KORDER_MIN=$k timeout 600 python tools/fwd_layer_times.py fwd dgrad --clips=$c --only=s2.p0,s3.p0,s4.p0,s5.p0 > $OUT/layers_${c}_korder$k.txt 2>&1; echo "layers $c korder $k exit $?"
  done
done
for rep in 1 2; do for k in 2 0; do
  VS_CONV_KORDER_MIN=$k timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-feat-fwd 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train korder_min $k', d['value'], d['ms_per_step'])"
  VS_CONV_KORDER_MIN=$k timeout 600 python bench.py --workload feat_fwd --steps 50 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fwd korder_min $k', d['value'], d['ms_per_step'])"
done; done
bash tools/timeline.sh sf_txenc_train > $OUT/timeline_train.txt 2>&1; tail -60 $OUT/timeline_train.txt
