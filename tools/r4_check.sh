#!/bin/bash
# Round-4: layer-local parity table with the chunk-major / tap-major reduction order; full GPU suite (no -x); timeline.
export TMPDIR=/tmp
OUT=gpurun_out/r4_check; mkdir -p $OUT
for k in 2 0; do
  VS_CONV_KORDER_MIN=$k timeout 900 python -m pytest "tests/test_gpu_parity_full.py" -q -m gpu -s -k "every_resblock and 64" --no-header -p no:cacheprovider > $OUT/parity_blocks_korder$k.log 2>&1; echo "parity blocks korder $k exit $?"; grep -E "^  [0-9]" $OUT/parity_blocks_korder$k.log | head -6
done
timeout 3000 python -m pytest tests -q -m gpu --no-header -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -5 $OUT/pytest_gpu.log
bash tools/timeline.sh sf_txenc_train > $OUT/timeline_train.txt 2>&1; tail -16 $OUT/timeline_train.txt
timeout 900 python bench.py > $OUT/bench_train.json 2> $OUT/bench_train.err; echo "bench exit $?"; python -c "
import json; d=json.loads(open('$OUT/bench_train.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('feat_fwd'), d['roofline']['all_conv'], d['roofline']['frac'], d['roofline']['kernel'])"
