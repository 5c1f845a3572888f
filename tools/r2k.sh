#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r2k; mkdir -p $OUT
run() { name=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 5 > $OUT/$name.json 2> $OUT/$name.err; echo -n "$name: "; python -c "import json,sys; print(json.load(open('$OUT/$name.json'))['ms_per_step'])" 2>/dev/null || (echo fail; tail -3 $OUT/$name.err); }
run lag0 VS_WGRAD_LAG=0
run lag1 VS_WGRAD_LAG=1
run lag2 VS_WGRAD_LAG=2
run lag0b VS_WGRAD_LAG=0
run lag1b VS_WGRAD_LAG=1
timeout 600 python -m pytest tests/test_gpu_trunk.py -q -m gpu --no-header -p no:cacheprovider -k "hipgraph or trajectory" > $OUT/pytest.log 2>&1; echo "pytest lag0 exit $?"; tail -2 $OUT/pytest.log
VS_WGRAD_LAG=1 timeout 600 python -m pytest tests/test_gpu_trunk.py -q -m gpu --no-header -p no:cacheprovider -k "hipgraph or trajectory" > $OUT/pytest_lag1.log 2>&1; echo "pytest lag1 exit $?"; tail -2 $OUT/pytest_lag1.log
timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_labels.json 2> $OUT/bench_labels.err; python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r2k/bench_labels.json") if l.startswith("{")][0])
for f in d["roofline"]["families"][:10]: print(f["kernel"][:70], f["ms_per_step"], f.get("frac"), f.get("traffic"))
PY
