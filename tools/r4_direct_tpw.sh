#!/bin/bash
# Round-4: small-channel direct kernel, tiles per wave chosen for >= N blocks (VS_DIRECT_TPW_BLOCKS = 0 (always 8) / 512 / 1024).
export TMPDIR=/tmp
OUT=gpurun_out/r4_direct_tpw; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_conv.py -q -m gpu -x -k "direct or small_channel or fused_bc" --no-header -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -2 $OUT/pytest.log
for b in 0 512 1024 0 512; do
  VS_DIRECT_TPW_BLOCKS=$b timeout 900 python tools/fwd_layer_times.py fwd dgrad --only=p1,fuse --small > $OUT/small_$b.txt 2>&1; echo "blocks $b: $(tail -2 $OUT/small_$b.txt | tr '\n' ' ')"
done
for rep in 1 2; do for b in 512 0; do
  VS_DIRECT_TPW_BLOCKS=$b timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-feat-fwd 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train tpw_blocks $b', d['value'], d['ms_per_step'])"
  VS_DIRECT_TPW_BLOCKS=$b timeout 600 python bench.py --workload feat_fwd --steps 50 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fwd tpw_blocks $b', d['value'], d['ms_per_step'])"
done; done
