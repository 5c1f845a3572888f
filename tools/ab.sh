#!/bin/bash
# The one-purpose GPU-session scripts of rounds 2-4 (tools/r2_*.sh, r3_*.sh, r4_*.sh until round 5), folded into one file:
#   tools/ab.sh <experiment> [args...]      e.g.  tools/ab.sh r4_whatif     tools/ab.sh r4_ab_lib OLD.so NEW.so
# profiles/*.txt cite them by their old names: `tools/r4_whatif.sh` == `tools/ab.sh r4_whatif`.  Each body is the old
# script verbatim (its own comment first).  Alternating A/B of ONE environment switch: tools/ab_env.sh.
# The what-if experiments skip kernel launches (garbage numerics): they set VS_WHATIF_OK=1, the guard ops.py asks for.
exp="$1"; shift
case "$exp" in
r2_final)
# Round-2 artefact run: full GPU test suite, smoke, both bench workloads (default flags), one-stream rocprofv3
# kernel stats, PMC traffic and MFMA-busy passes.  Everything lands in gpurun_out/r2_final/.
export TMPDIR=/tmp
TAG=${1:-r02}
OUT=gpurun_out/r2_final; mkdir -p $OUT
export VS_BUILD_TAG="$TAG"
timeout 2400 python -m pytest tests -q -m gpu --no-header -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
timeout 600 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -3 $OUT/smoke.log
( time timeout 900 python bench.py ) > $OUT/bench_train.json 2> $OUT/bench_train.err; echo "bench train exit $?"; head -c 300 $OUT/bench_train.json; echo; tail -4 $OUT/bench_train.err
( time timeout 600 python bench.py --workload feat_fwd ) > $OUT/bench_feat_fwd.json 2> $OUT/bench_feat_fwd.err; echo "bench fwd exit $?"; head -c 300 $OUT/bench_feat_fwd.json; echo
export VS_DUAL_STREAM=0 VS_WGRAD_LANES=0 VS_CONV_PAIR=0
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train -- python3 bench.py --steps 5 --warmup 2 --workload sf_txenc_train --no-cpu-baseline --no-roofline --graph 0 > $OUT/rocprof_train.log 2>&1; echo "rocprof train exit $?"
f=$(find $OUT/prof_train -name "*kernel_stats*.csv" | head -1); cp "$f" $OUT/train_kernel_stats_one_stream.csv; head -12 "$f" | cut -c1-150
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_fwd -- python3 bench.py --steps 5 --warmup 2 --workload feat_fwd --no-cpu-baseline --no-roofline --graph 0 > $OUT/rocprof_fwd.log 2>&1; echo "rocprof fwd exit $?"
f=$(find $OUT/prof_fwd -name "*kernel_stats*.csv" | head -1); cp "$f" $OUT/feat_fwd_kernel_stats_one_stream.csv
find $OUT -name "*kernel_trace*.csv" -delete
unset VS_DUAL_STREAM VS_WGRAD_LANES VS_CONV_PAIR
bash tools/pmc_traffic.sh sf_txenc_train > $OUT/pmc_traffic.log 2>&1; tail -12 $OUT/pmc_traffic.log; cp gpurun_out/pmc_traffic/pmc_traffic.json $OUT/pmc_traffic.json
bash tools/pmc_mfma.sh > $OUT/pmc_mfma.log 2>&1; tail -28 $OUT/pmc_mfma.log; cp gpurun_out/pmc_mfma/pmc_mfma.json $OUT/pmc_mfma.json
;;
r3_ab_noslp)
export TMPDIR=/tmp
OUT=gpurun_out/r3_d; mkdir -p $OUT
export VS_WGRAD_S1_TILES=128
for i in 1 2 3; do
  for lib in "" "/root/repo/tmp/noslp/libvidsitu_hip.so"; do
    env VS_LIB_PATH=$lib timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train lib=${lib:-default}', d['value'], d['ms_per_step'])" | tee -a $OUT/noslp_ab.log
  done
done
for lib in "" "/root/repo/tmp/noslp/libvidsitu_hip.so"; do
  env VS_LIB_PATH=$lib timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 200 --workload feat_fwd 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fwd lib=${lib:-default}', d['value'], d['ms_per_step'])" | tee -a $OUT/noslp_ab.log
done
;;
r3_ab_wgrad_unsplit)
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
;;
r3_batch_probe)
# batch-size probe: how the same kernels fill the chip at 16 / 32 clips per GPU (not the headline config)
mkdir -p gpurun_out/r3_i
for wl in sf_txenc_train feat_fwd; do
for n in 8 16 32; do
  timeout 600 python bench.py --workload $wl --clips-per-gpu $n --steps 20 --warmup 5 --no-cpu-baseline --no-roofline \
    > gpurun_out/r3_i/${wl}_$n.json 2> gpurun_out/r3_i/${wl}_$n.err
  echo "$wl $n exit $?"; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r3_i/${wl}_$n.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["config"].get("frac_of_bf16_mfma_peak"))
except Exception as e:
    print("no line", e)
PY
done; done
;;
r3_batch_probe_large)
timeout 1500 python -m pytest tests/test_gpu_conv.py -q -m gpu 2>&1 | tail -8
;;
r3_final)
# Round-2 artefact run: full GPU test suite, smoke, both bench workloads (default flags), one-stream rocprofv3
# kernel stats, PMC traffic and MFMA-busy passes.  Everything lands in gpurun_out/r3_final/.
export TMPDIR=/tmp
TAG=${1:-r03}
OUT=gpurun_out/r3_final; mkdir -p $OUT
export VS_BUILD_TAG="$TAG"
timeout 2400 python -m pytest tests -q -m gpu --no-header -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
timeout 600 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -3 $OUT/smoke.log
( time timeout 900 python bench.py ) > $OUT/bench_train.json 2> $OUT/bench_train.err; echo "bench train exit $?"; head -c 300 $OUT/bench_train.json; echo; tail -4 $OUT/bench_train.err
( time timeout 600 python bench.py --workload feat_fwd ) > $OUT/bench_feat_fwd.json 2> $OUT/bench_feat_fwd.err; echo "bench fwd exit $?"; head -c 300 $OUT/bench_feat_fwd.json; echo
export VS_DUAL_STREAM=0 VS_WGRAD_LANES=0 VS_CONV_PAIR=0
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train -- python3 bench.py --steps 5 --warmup 2 --workload sf_txenc_train --no-cpu-baseline --no-roofline --graph 0 > $OUT/rocprof_train.log 2>&1; echo "rocprof train exit $?"
f=$(find $OUT/prof_train -name "*kernel_stats*.csv" | head -1); cp "$f" $OUT/train_kernel_stats_one_stream.csv; head -12 "$f" | cut -c1-150
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_fwd -- python3 bench.py --steps 5 --warmup 2 --workload feat_fwd --no-cpu-baseline --no-roofline --graph 0 > $OUT/rocprof_fwd.log 2>&1; echo "rocprof fwd exit $?"
f=$(find $OUT/prof_fwd -name "*kernel_stats*.csv" | head -1); cp "$f" $OUT/feat_fwd_kernel_stats_one_stream.csv
find $OUT -name "*kernel_trace*.csv" -delete
unset VS_DUAL_STREAM VS_WGRAD_LANES VS_CONV_PAIR
bash tools/pmc_traffic.sh sf_txenc_train > $OUT/pmc_traffic.log 2>&1; tail -12 $OUT/pmc_traffic.log; cp gpurun_out/pmc_traffic/pmc_traffic.json $OUT/pmc_traffic.json
bash tools/pmc_mfma.sh > $OUT/pmc_mfma.log 2>&1; tail -28 $OUT/pmc_mfma.log; cp gpurun_out/pmc_mfma/pmc_mfma.json $OUT/pmc_mfma.json
;;
r3_parity)
export TMPDIR=/tmp
OUT=gpurun_out/r3_parity; mkdir -p $OUT
timeout 3000 python -m pytest tests/test_gpu_parity_full.py tests/test_gpu_bn_pool.py tests/test_gpu_train_step.py -q -m gpu --no-header -p no:cacheprovider -s > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest.log
grep -E "relative|rel_l2|passed|failed|FAILED|Error" $OUT/pytest.log | cut -c1-400 | head -60
for v in 0 1; do
  env VS_RESIDUAL_FP32=$v timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 100 --workload feat_fwd 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fwd VS_RESIDUAL_FP32=$v', d['value'], d['ms_per_step'])" | tee -a $OUT/ab.log
done
;;
r4_ab_deep)
# Round-4: full GPU suite on the build with the deep-pipeline kernel, then the train step and the eval forward with the
# kernel's plan on / off (VS_CONV_DEEP=1 / 0), alternating in one session; 8 and 32 clips per GPU.
export TMPDIR=/tmp
OUT=gpurun_out/r4_ab_deep; mkdir -p $OUT
python - <<'PY' > $OUT/visible_gpus.txt 2>&1
from vidsitu_amd import dist_launch
import glob
print("kfd nodes:", len(glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")), "visible_gpus:", dist_launch.visible_gpus())
PY
cat $OUT/visible_gpus.txt
timeout 3000 python -m pytest tests -q -m gpu --no-header -p no:cacheprovider -x > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log; tail -5 $OUT/pytest_gpu.log
cp gpurun_out/parity_eval.json $OUT/ 2>/dev/null
for rep in 1 2; do
for deep in 1 0; do
for c in 8 32; do
  VS_CONV_DEEP=$deep timeout 600 python bench.py --workload sf_txenc_train --clips-per-gpu $c --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-feat-fwd > $OUT/train_${c}_deep${deep}_$rep.json 2> $OUT/train_${c}_deep${deep}_$rep.err
  VS_CONV_DEEP=$deep timeout 600 python bench.py --workload feat_fwd --clips-per-gpu $c --steps 50 --warmup 5 --no-cpu-baseline --no-roofline > $OUT/fwd_${c}_deep${deep}_$rep.json 2> $OUT/fwd_${c}_deep${deep}_$rep.err
  python - <<PY
import json
for wl in ("train","fwd"):
    try:
        d=json.loads(open("$OUT/%s_${c}_deep${deep}_$rep.json"%wl).read().strip().splitlines()[-1]); print(wl,"clips",$c,"deep",$deep,"rep",$rep,d["value"],d["ms_per_step"])
    except Exception as e: print(wl,"no line",e)
PY
done; done; done
;;
r4_ab_lib)
# A/B of two builds of the library in one session (alternating): bash tools/ab.sh r4_ab_lib <other .so> [clips ...]
# The train step and the eval forward with the in-tree library ("new") and with VS_LIB_PATH=<other> ("old").
export TMPDIR=/tmp
OTHER=$1; shift; CL=${@:-8}
OUT=gpurun_out/r4_ab_lib; mkdir -p $OUT
for rep in 1 2 3; do
for which in new old; do
for c in $CL; do
  if [ $which = old ]; then export VS_LIB_PATH=$OTHER; else unset VS_LIB_PATH; fi
  timeout 600 python bench.py --workload sf_txenc_train --clips-per-gpu $c --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-feat-fwd > $OUT/train_${c}_${which}_$rep.json 2> $OUT/train_${c}_${which}_$rep.err
  timeout 600 python bench.py --workload feat_fwd --clips-per-gpu $c --steps 50 --warmup 5 --no-cpu-baseline --no-roofline > $OUT/fwd_${c}_${which}_$rep.json 2> $OUT/fwd_${c}_${which}_$rep.err
  python - <<PY
import json
for wl in ("train","fwd"):
    try:
        d=json.loads(open("$OUT/%s_${c}_${which}_$rep.json"%wl).read().strip().splitlines()[-1]); print(wl,"clips",$c,"$which","rep",$rep,d["value"],d["ms_per_step"])
    except Exception as e: print(wl,"no line",e)
PY
done; done; done
;;
r4_baseline_tables)
# Round-4 first GPU session: the fill-path ceiling (LDS-DMA rate per CU by serving level, waves and depth) and the
# per-layer tables of the round-3 build at 8 clips (the bench shape) and 32 clips (each kernel's steady state).
export TMPDIR=/tmp
OUT=gpurun_out/r4_baseline; mkdir -p $OUT
timeout 300 tools/probes/lds_dma_rate > $OUT/lds_dma_rate.txt 2>&1; echo "dma probe exit $?"; head -30 $OUT/lds_dma_rate.txt
timeout 900 python tools/fwd_layer_times.py fwd dgrad wgrad > $OUT/layer_times_8.txt 2>&1; echo "layers8 exit $?"; tail -4 $OUT/layer_times_8.txt
timeout 900 python tools/fwd_layer_times.py fwd dgrad wgrad --clips=32 --only=s2.p0,s3.p0,s4.p0,s5.p0 > $OUT/layer_times_32.txt 2>&1; echo "layers32 exit $?"; tail -4 $OUT/layer_times_32.txt
;;
r4_batch_probe)
# Round-4 batch probe: the same step at 8 / 16 / 32 / 40 / 64 clips per GPU (8 = the BASELINE config; 40 = the reference's
# default train.bs x 5 events; the others are probes), deep kernels on / off.
export TMPDIR=/tmp
for n in 8 16 32 40 64; do for deep in 1 0; do
  for wl in sf_txenc_train feat_fwd; do
    VS_CONV_DEEP=$deep VS_WGRAD_DEEP=$deep timeout 600 python bench.py --workload $wl --clips-per-gpu $n --steps 15 --warmup 4 --no-cpu-baseline --no-roofline --no-feat-fwd 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl clips $n deep $deep', d['value'], d['ms_per_step'], d['config'].get('frac_of_bf16_mfma_peak'))"
  done
done; done
;;
r4_bnred_ab)
# BN reduce kernels with predicated 16-load rounds: BN / trunk / train-step tests, then the step A/B against the previous build
export TMPDIR=/tmp
OUT=gpurun_out/r4_bnred; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_bn_pool.py tests/test_gpu_trunk.py tests/test_gpu_train_step.py -q -m gpu --no-header -p no:cacheprovider -x > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest.log; tail -3 $OUT/pytest.log
bash tools/ab.sh r4_ab_lib $1 8
;;
r4_check)
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
;;
r4_deep_first)
# Round-4: first GPU session of the deep-pipeline conv kernel -- its parity tests, then the per-layer tables with the
# kernel forced on / off at 8 and 32 clips (slow pathway, MFMA-side layers).
export TMPDIR=/tmp
OUT=gpurun_out/r4_deep_first; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_conv.py -q -m gpu -x -k "deep_pipeline" --no-header -p no:cacheprovider > $OUT/pytest_deep.log 2>&1; echo "pytest deep exit $?"; tail -15 $OUT/pytest_deep.log
for c in 8 32; do
  timeout 600 python tools/fwd_layer_times.py fwd dgrad --clips=$c --only=s3.p0,s4.p0,s5.p0 --deep > $OUT/layers_${c}_deep.txt 2>&1; echo "deep $c exit $?"
  timeout 600 python tools/fwd_layer_times.py fwd dgrad --clips=$c --only=s3.p0,s4.p0,s5.p0 --nodeep > $OUT/layers_${c}_nodeep.txt 2>&1; echo "nodeep $c exit $?"
done
tail -30 $OUT/layers_8_deep.txt | cut -c1-120
;;
r4_direct_tpw)
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
;;
r5_batch_probe)
# Round-5 batch probe: the step at 8 / 16 / 32 / 40 / 64 clips per GPU on the default plan (8 = the BASELINE config, 40 = the
# reference's train.bs x 5 events); a least-squares line t = a + b * clips through the train rows is printed at the end.
export TMPDIR=/tmp
OUT=gpurun_out/r5_batch_probe; mkdir -p $OUT; : > $OUT/probe.txt
for n in 8 16 32 40 64; do
  for wl in sf_txenc_train feat_fwd; do
    timeout 600 python bench.py --workload $wl --clips-per-gpu $n --steps 15 --warmup 4 --no-cpu-baseline --no-roofline --no-feat-fwd 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl clips $n', d['value'], d['ms_per_step'], d['config'].get('frac_of_bf16_mfma_peak'))" | tee -a $OUT/probe.txt
  done
done
python - <<'PY' | tee -a gpurun_out/r5_batch_probe/probe.txt
import numpy as np
rows = [l.split() for l in open("gpurun_out/r5_batch_probe/probe.txt") if l.startswith("sf_txenc_train")]
x = np.array([float(r[2]) for r in rows]); y = np.array([float(r[4]) for r in rows])
for sel, name in ((x >= 16, "16..64"), (x >= 0, "8..64")):
    b, a = np.polyfit(x[sel], y[sel], 1)
    print(f"train, clips {name}: t = {a:.2f} ms + {b:.3f} ms/clip; the 8-clip step is {y[0]:.2f} ms, {y[0] - 8 * b:.2f} ms of it do not scale with the batch")
PY
;;
r6_final)
# Round-6 artefact run: full GPU test suite, smoke, both bench workloads (default flags), one-stream rocprofv3
# kernel stats, PMC traffic and MFMA-busy passes, the batch probe.  Everything lands in gpurun_out/r6_final/.
export TMPDIR=/tmp
TAG=${1:-r06}
OUT=gpurun_out/r6_final; mkdir -p $OUT
export VS_BUILD_TAG="$TAG"
timeout 2700 python -m pytest tests -q -m gpu --no-header -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
timeout 600 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -3 $OUT/smoke.log
( time timeout 900 python bench.py ) > $OUT/bench_train.json 2> $OUT/bench_train.err; echo "bench train exit $?"; head -c 300 $OUT/bench_train.json; echo; tail -4 $OUT/bench_train.err
( time timeout 600 python bench.py --workload feat_fwd ) > $OUT/bench_feat_fwd.json 2> $OUT/bench_feat_fwd.err; echo "bench fwd exit $?"; head -c 300 $OUT/bench_feat_fwd.json; echo
export VS_DUAL_STREAM=0 VS_WGRAD_LANES=0 VS_CONV_PAIR=0
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train -- python3 bench.py --steps 5 --warmup 2 --workload sf_txenc_train --no-cpu-baseline --no-roofline --graph 0 > $OUT/rocprof_train.log 2>&1; echo "rocprof train exit $?"
f=$(find $OUT/prof_train -name "*kernel_stats*.csv" | head -1); cp "$f" $OUT/train_kernel_stats_one_stream.csv; head -12 "$f" | cut -c1-150
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_fwd -- python3 bench.py --steps 5 --warmup 2 --workload feat_fwd --no-cpu-baseline --no-roofline --graph 0 > $OUT/rocprof_fwd.log 2>&1; echo "rocprof fwd exit $?"
f=$(find $OUT/prof_fwd -name "*kernel_stats*.csv" | head -1); cp "$f" $OUT/feat_fwd_kernel_stats_one_stream.csv
find $OUT -name "*kernel_trace*.csv" -delete
unset VS_DUAL_STREAM VS_WGRAD_LANES VS_CONV_PAIR
bash tools/pmc_traffic.sh sf_txenc_train > $OUT/pmc_traffic.log 2>&1; tail -12 $OUT/pmc_traffic.log; cp gpurun_out/pmc_traffic/pmc_traffic.json $OUT/pmc_traffic.json
bash tools/pmc_traffic.sh feat_fwd > $OUT/pmc_traffic_fwd.log 2>&1; cp gpurun_out/pmc_traffic/pmc_traffic.json $OUT/pmc_traffic_feat_fwd.json
bash tools/pmc_mfma.sh > $OUT/pmc_mfma.log 2>&1; tail -28 $OUT/pmc_mfma.log; cp gpurun_out/pmc_mfma/pmc_mfma.json $OUT/pmc_mfma.json
bash tools/ab.sh r5_batch_probe > $OUT/batch_probe.log 2>&1; cp gpurun_out/r5_batch_probe/probe.txt $OUT/batch_probe.txt; tail -3 $OUT/batch_probe.txt
;;
r5_final)
# Round-5 artefact run: full GPU test suite, smoke, both bench workloads (default flags), one-stream rocprofv3
# kernel stats, PMC traffic and MFMA-busy passes.  Everything lands in gpurun_out/r5_final/.
export TMPDIR=/tmp
TAG=${1:-r05}
OUT=gpurun_out/r5_final; mkdir -p $OUT
export VS_BUILD_TAG="$TAG"
timeout 2400 python -m pytest tests -q -m gpu --no-header -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
timeout 600 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -3 $OUT/smoke.log
( time timeout 900 python bench.py ) > $OUT/bench_train.json 2> $OUT/bench_train.err; echo "bench train exit $?"; head -c 300 $OUT/bench_train.json; echo; tail -4 $OUT/bench_train.err
( time timeout 600 python bench.py --workload feat_fwd ) > $OUT/bench_feat_fwd.json 2> $OUT/bench_feat_fwd.err; echo "bench fwd exit $?"; head -c 300 $OUT/bench_feat_fwd.json; echo
export VS_DUAL_STREAM=0 VS_WGRAD_LANES=0 VS_CONV_PAIR=0
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train -- python3 bench.py --steps 5 --warmup 2 --workload sf_txenc_train --no-cpu-baseline --no-roofline --graph 0 > $OUT/rocprof_train.log 2>&1; echo "rocprof train exit $?"
f=$(find $OUT/prof_train -name "*kernel_stats*.csv" | head -1); cp "$f" $OUT/train_kernel_stats_one_stream.csv; head -12 "$f" | cut -c1-150
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_fwd -- python3 bench.py --steps 5 --warmup 2 --workload feat_fwd --no-cpu-baseline --no-roofline --graph 0 > $OUT/rocprof_fwd.log 2>&1; echo "rocprof fwd exit $?"
f=$(find $OUT/prof_fwd -name "*kernel_stats*.csv" | head -1); cp "$f" $OUT/feat_fwd_kernel_stats_one_stream.csv
find $OUT -name "*kernel_trace*.csv" -delete
unset VS_DUAL_STREAM VS_WGRAD_LANES VS_CONV_PAIR
bash tools/pmc_traffic.sh sf_txenc_train > $OUT/pmc_traffic.log 2>&1; tail -12 $OUT/pmc_traffic.log; cp gpurun_out/pmc_traffic/pmc_traffic.json $OUT/pmc_traffic.json
bash tools/pmc_mfma.sh > $OUT/pmc_mfma.log 2>&1; tail -28 $OUT/pmc_mfma.log; cp gpurun_out/pmc_mfma/pmc_mfma.json $OUT/pmc_mfma.json
;;
r4_final)
# Round-4 artefact run: full GPU test suite, smoke, both bench workloads (default flags), one-stream rocprofv3
# kernel stats, PMC traffic and MFMA-busy passes.  Everything lands in gpurun_out/r4_final/.
export TMPDIR=/tmp
TAG=${1:-r04}
OUT=gpurun_out/r4_final; mkdir -p $OUT
export VS_BUILD_TAG="$TAG"
timeout 2400 python -m pytest tests -q -m gpu --no-header -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
timeout 600 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -3 $OUT/smoke.log
( time timeout 900 python bench.py ) > $OUT/bench_train.json 2> $OUT/bench_train.err; echo "bench train exit $?"; head -c 300 $OUT/bench_train.json; echo; tail -4 $OUT/bench_train.err
( time timeout 600 python bench.py --workload feat_fwd ) > $OUT/bench_feat_fwd.json 2> $OUT/bench_feat_fwd.err; echo "bench fwd exit $?"; head -c 300 $OUT/bench_feat_fwd.json; echo
export VS_DUAL_STREAM=0 VS_WGRAD_LANES=0 VS_CONV_PAIR=0
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train -- python3 bench.py --steps 5 --warmup 2 --workload sf_txenc_train --no-cpu-baseline --no-roofline --graph 0 > $OUT/rocprof_train.log 2>&1; echo "rocprof train exit $?"
f=$(find $OUT/prof_train -name "*kernel_stats*.csv" | head -1); cp "$f" $OUT/train_kernel_stats_one_stream.csv; head -12 "$f" | cut -c1-150
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_fwd -- python3 bench.py --steps 5 --warmup 2 --workload feat_fwd --no-cpu-baseline --no-roofline --graph 0 > $OUT/rocprof_fwd.log 2>&1; echo "rocprof fwd exit $?"
f=$(find $OUT/prof_fwd -name "*kernel_stats*.csv" | head -1); cp "$f" $OUT/feat_fwd_kernel_stats_one_stream.csv
find $OUT -name "*kernel_trace*.csv" -delete
unset VS_DUAL_STREAM VS_WGRAD_LANES VS_CONV_PAIR
bash tools/pmc_traffic.sh sf_txenc_train > $OUT/pmc_traffic.log 2>&1; tail -12 $OUT/pmc_traffic.log; cp gpurun_out/pmc_traffic/pmc_traffic.json $OUT/pmc_traffic.json
bash tools/pmc_mfma.sh > $OUT/pmc_mfma.log 2>&1; tail -28 $OUT/pmc_mfma.log; cp gpurun_out/pmc_mfma/pmc_mfma.json $OUT/pmc_mfma.json
;;
r4_halo_pmc)
# Round-4: full GPU suite; halo planner's conflict price A/B (padded lines vs conflicting fragment reads);
# matrix-core busy / stall / LDS-conflict counters incl. the launches of the deep-pipeline kernels.
export TMPDIR=/tmp
OUT=gpurun_out/r4_halo_pmc; mkdir -p $OUT
timeout 3000 python -m pytest tests -q -m gpu --no-header -p no:cacheprovider -x > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest_gpu.log
for wgt in 0.85 0.6; do
  VS_HALO_CONFLICT_WEIGHT=$wgt timeout 600 python tools/fwd_layer_times.py fwd dgrad --only=s4.p0.b1.b,s5.p0.b1.b,s4.p0.b0.b,s5.p0.b0.b,s3.p0.b1.b > $OUT/halo_w$wgt.txt 2>&1; echo "halo weight $wgt:"; grep "^s" $OUT/halo_w$wgt.txt | cut -c1-110
done
VS_BUILD_TAG=r04-v1 bash tools/pmc_mfma.sh > $OUT/pmc_mfma.log 2>&1; tail -45 $OUT/pmc_mfma.log; cp gpurun_out/pmc_mfma/pmc_mfma.json $OUT/pmc_mfma.json
;;
r4_knob_sweep)
# Round-4, late build: every shipped switch / plan knob alone against the default in the train step, alternating, one GPU session
export TMPDIR=/tmp
OUT=gpurun_out/r4_knobs; mkdir -p $OUT; F=$OUT/knobs.txt; : > $F
run() { env $1 timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-feat-fwd --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-30s %8.2f clips/s %7.3f ms' % ('$1', d['value'], d['ms_per_step']))" | tee -a $F; }
for i in 1 2; do
  for cfg in DEFAULT=1 VS_CONV_DEEP=2 VS_CONV_DEEP=0 VS_CONV_KORDER_MIN=0 VS_WGRAD_DEEP_STAG=0 VS_TRAIN_AOL=1 \
             DEFAULT=2 VS_DIRECT_BNB=1 VS_WGRAD_XCD=0 VS_CONV_HALO=2 VS_CONV_HALO=0 VS_HALO_CONFLICT_WEIGHT=0.6 VS_REDUCE_MERGE=0 VS_CONV_PAIR=0 \
             DEFAULT=3 VS_BN_TWO_LEVEL=1600 VS_BN_TWO_LEVEL=256 VS_WGRAD_LANES=0 VS_ADAM_OVERLAP=1 VS_WGRAD_SLOTS_SMALL=256 VS_WGRAD_SLOTS_SMALL=1024 VS_DIRECT_TB=2; do run $cfg; done
done
;;
r4_korder_timeline)
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
;;
r4_kstats)
# one-stream rocprofv3 kernel stats of the train step (eager, 5 steps) -> gpurun_out/r4_kstats/train_kernel_stats.csv
export TMPDIR=/tmp
OUT=gpurun_out/r4_kstats; mkdir -p $OUT
export VS_DUAL_STREAM=0 VS_WGRAD_LANES=0 VS_CONV_PAIR=0
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train -- python3 bench.py --steps 5 --warmup 2 --workload sf_txenc_train --no-cpu-baseline --no-roofline --no-feat-fwd --graph 0 > $OUT/rocprof_train.log 2>&1; echo "rocprof train exit $?"
f=$(find $OUT/prof_train -name "*kernel_stats*.csv" | head -1); cp "$f" $OUT/train_kernel_stats.csv; head -14 "$f" | cut -c1-160
find $OUT -name "*kernel_trace*.csv" -delete
;;
r4_kstats32)
export TMPDIR=/tmp
OUT=gpurun_out/r4_kstats32; mkdir -p $OUT
export VS_DUAL_STREAM=0 VS_WGRAD_LANES=0 VS_CONV_PAIR=0
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --steps 4 --warmup 2 --clips-per-gpu 32 --workload sf_txenc_train --no-cpu-baseline --no-roofline --no-feat-fwd --graph 0 > $OUT/log.txt 2>&1; echo exit $?
f=$(find $OUT/prof -name "*kernel_stats*.csv" | head -1); cp "$f" $OUT/train32_kernel_stats.csv; find $OUT -name "*kernel_trace*.csv" -delete
;;
r4_nt_ab)
# non-temporal hints: optimizer / train-step / txenc tests, then the step A/B against the previous build
export TMPDIR=/tmp
OUT=gpurun_out/r4_nt; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_txenc.py tests/test_gpu_train_step.py tests/test_gpu_trunk.py tests/test_gpu_bn_pool.py -q -m gpu --no-header -p no:cacheprovider -x > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest.log; tail -3 $OUT/pytest.log
bash tools/ab.sh r4_ab_lib $1 8
;;
r4_plan_ab)
# conv tests + trunk tests on the new plan rules, then the step A/B against the previous build
export TMPDIR=/tmp
OUT=gpurun_out/r4_plan; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_conv.py tests/test_gpu_trunk.py -q -m gpu --no-header -p no:cacheprovider -x > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest.log; tail -4 $OUT/pytest.log
bash tools/ab.sh r4_ab_lib $1 8
;;
r4_reduce_ab)
# Round-4: the one-thread-per-column slab reduce (S <= 16): weight-gradient tests (bitwise vs the batched slice-form kernel), then the
# step A/B against the previous build (bash tools/ab.sh r4_ab_lib).
export TMPDIR=/tmp
OUT=gpurun_out/r4_reduce; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_conv.py tests/test_gpu_trunk.py tests/test_gpu_train_step.py -q -m gpu --no-header -p no:cacheprovider -x -k "wgrad or reduce or grad or step or pair" > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest.log; tail -4 $OUT/pytest.log
bash tools/ab.sh r4_ab_lib $1 8
;;
r4_splitk_il)
# Round-4: the in-launch split-K plan of the 128 x 128 tile kernel: its tests, then per-layer times of the slow pathway at
# 8 clips with the plan off / heuristic / forced, with and without the halo-image and deep kernels in front of it.
export TMPDIR=/tmp
OUT=gpurun_out/r4_splitk_il; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_conv.py -q -m gpu --no-header -p no:cacheprovider -x -k "splitk or deep_pipeline or halo_image" > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest.log; tail -8 $OUT/pytest.log
ONLY=--only=s3.p0,s4.p0,s5.p0,_fuse
for il in 0 1 2; do
  VS_CONV_SPLITK_IL=$il timeout 600 python tools/fwd_layer_times.py fwd dgrad --clips=8 $ONLY > $OUT/plan_il$il.txt 2>&1
  echo "== default plan order, VS_CONV_SPLITK_IL=$il"; tail -32 $OUT/plan_il$il.txt | cut -c1-120
done
for il in 0 2; do
  VS_CONV_SPLITK_IL=$il timeout 600 python tools/fwd_layer_times.py fwd dgrad --clips=8 --nohalo --nodeep $ONLY > $OUT/tile_il$il.txt 2>&1
  echo "== tile kernel only (--nohalo --nodeep), VS_CONV_SPLITK_IL=$il"; tail -32 $OUT/tile_il$il.txt | cut -c1-120
done
;;
r4_wgrad_align_probe)
# Round-4: (1) deep weight-gradient kernel with whole splits per XCD (VS_WGRAD_DEEP_ALIGN=1) vs not, 32 and 8 clips;
# (2) the GEMM probe's staggered all-wave copy issue against halves-burst.
export TMPDIR=/tmp
OUT=gpurun_out/r4_wgrad_align; mkdir -p $OUT
for c in 32 8; do for al in 1 0; do
  VS_WGRAD_DEEP_ALIGN=$al timeout 900 python tools/fwd_layer_times.py wgrad --clips=$c --only=s3.p0,s4.p0,s5.p0 --deep > $OUT/wgrad_${c}_align$al.txt 2>&1; echo "clips $c align $al: $(tail -1 $OUT/wgrad_${c}_align$al.txt)"
done; done
timeout 600 tools/probes/gemm_deep > $OUT/gemm_deep_v4.txt 2>&1; echo "probe exit $?"; cut -c1-250 $OUT/gemm_deep_v4.txt | grep -v check
;;
r4_wgrad_deep)
# Round-4: the deep-pipeline weight-gradient kernel -- parity tests, per-layer table forced / plan / off at 8 and 32 clips,
# the train step with the plan on / off (VS_WGRAD_DEEP=1 / 0).
export TMPDIR=/tmp
OUT=gpurun_out/r4_wgrad_deep; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_conv.py -q -m gpu -x -k "wgrad" --no-header -p no:cacheprovider > $OUT/pytest_wgrad.log 2>&1; echo "pytest wgrad exit $?"; tail -12 $OUT/pytest_wgrad.log
for c in 8 32; do
  only=""; [ $c = 32 ] && only="--only=s2.p0,s3.p0,s4.p0,s5.p0"
  timeout 900 python tools/fwd_layer_times.py wgrad --clips=$c $only --deep > $OUT/wgrad_${c}_force.txt 2>&1; echo "force $c exit $?"; tail -1 $OUT/wgrad_${c}_force.txt
  timeout 900 python tools/fwd_layer_times.py wgrad --clips=$c $only > $OUT/wgrad_${c}_plan.txt 2>&1; echo "plan $c exit $?"; tail -1 $OUT/wgrad_${c}_plan.txt
  timeout 900 python tools/fwd_layer_times.py wgrad --clips=$c $only --nodeep > $OUT/wgrad_${c}_off.txt 2>&1; echo "off $c exit $?"; tail -1 $OUT/wgrad_${c}_off.txt
done
for rep in 1 2 3; do for a in 1 0; do
  VS_WGRAD_DEEP=$a timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-feat-fwd 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train wgrad_deep $a', d['value'], d['ms_per_step'])"
done; done
;;
r4_wgrad_deep_step)
# Round-4: deep weight gradients in the step: pair launches on / off x deep plan on / off, alternating.
export TMPDIR=/tmp
for rep in 1 2; do for pair in 1 0; do for a in 1 0; do
  VS_CONV_PAIR=$pair VS_WGRAD_DEEP=$a timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-feat-fwd 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train pair $pair wgrad_deep $a', d['value'], d['ms_per_step'])"
done; done; done
for a in 1 0; do
  VS_WGRAD_DEEP=$a timeout 600 python bench.py --clips-per-gpu 32 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-feat-fwd 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train 32 clips wgrad_deep $a', d['value'], d['ms_per_step'])"
done
;;
r4_wgrad_half)
# Round-4: weight-gradient ring kernel with half stages (4 x 32 positions in 64 KiB, 3 steps in flight; VS_WGRAD_HALF=1)
# against the two-stage ring: parity tests with it on, per-layer table, 8 clips.
export TMPDIR=/tmp
OUT=gpurun_out/r4_wgrad_half; mkdir -p $OUT
VS_WGRAD_HALF=1 timeout 900 python -m pytest tests/test_gpu_conv.py -q -m gpu -x -k "wgrad" --no-header -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest half exit $?"; tail -3 $OUT/pytest.log
for h in 1 0 1 0; do
  VS_WGRAD_HALF=$h timeout 900 python tools/fwd_layer_times.py wgrad > $OUT/wgrad_8_half${h}_$RANDOM.txt 2>&1; echo "half $h: $(tail -1 $(ls -t $OUT/wgrad_8_half${h}_*.txt | head -1))"
done
;;
r4_wgrad_stag)
# Round-4: deep weight-gradient kernel, staggered all-wave copy issue (VS_WGRAD_DEEP_STAG=1) against the halves issue.
export TMPDIR=/tmp
OUT=gpurun_out/r4_wgrad_stag; mkdir -p $OUT
VS_WGRAD_DEEP_STAG=1 timeout 600 python -m pytest tests/test_gpu_conv.py -q -m gpu -x -k "deep_pipeline_wgrad" --no-header -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest stag exit $?"; tail -3 $OUT/pytest.log
for c in 32 8; do for st in 1 0; do
  VS_WGRAD_DEEP_STAG=$st timeout 900 python tools/fwd_layer_times.py wgrad --clips=$c --only=s3.p0,s4.p0,s5.p0 --deep > $OUT/wgrad_${c}_stag$st.txt 2>&1; echo "clips $c stag $st: $(tail -1 $OUT/wgrad_${c}_stag$st.txt)"
done; done
;;
r4_wgrad_whatif)
# Round-4: (1) the train step WITHOUT its weight gradients (VS_WHATIF=4, garbage dW: timing only) against the real step;
# (a packed run of all weight gradients on 1-4 streams in one hipGraph was tried and died in hipStreamEndCapture: profiles/r04_wgrad_whatif.txt)
export TMPDIR=/tmp
for rep in 1 2; do for w in 0 4; do
  VS_WHATIF_OK=1 VS_WHATIF=$w timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-feat-fwd 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train whatif $w', d['value'], d['ms_per_step'])"
done; done
;;
r4_wgrad_xcd)
# Round-4: weight gradients with one position split per XCD (S a multiple of 8 + XCD-contiguous block order) against
# the round-3 split choice (VS_WGRAD_ALIGN8=0), per layer and in the step; then the train step's timeline.
export TMPDIR=/tmp
OUT=gpurun_out/r4_wgrad_xcd; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_conv.py -q -m gpu -x -k "wgrad" --no-header -p no:cacheprovider > $OUT/pytest_wgrad.log 2>&1; echo "pytest wgrad exit $?"; tail -3 $OUT/pytest_wgrad.log
for a in 1 0; do
  VS_WGRAD_ALIGN8=$a timeout 900 python tools/fwd_layer_times.py wgrad > $OUT/wgrad_8_align$a.txt 2>&1; echo "wgrad table align $a exit $?"; tail -1 $OUT/wgrad_8_align$a.txt
done
VS_WGRAD_ALIGN8=1 VS_WGRAD_XCD=2 timeout 900 python tools/fwd_layer_times.py wgrad > $OUT/wgrad_8_align1_xcd2.txt 2>&1; tail -1 $OUT/wgrad_8_align1_xcd2.txt
for rep in 1 2 3; do for a in 1 0; do
  VS_WGRAD_ALIGN8=$a timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-feat-fwd 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train align8 $a', d['value'], d['ms_per_step'])"
done; done
bash tools/timeline.sh sf_txenc_train > $OUT/timeline_train.txt 2>&1; tail -70 $OUT/timeline_train.txt
;;
r4_whatif)
# Round-4 what-if table: the replayed train step with whole kernel families skipped (VS_WHATIF bits: 1 bn_finalize, 2 bn_bwd_finalize,
# 4 conv_wgrad + slab reduce, 8 bn_apply / bn_bwd_reduce / bn_bwd_apply).  Garbage numerics: timing only.
export TMPDIR=/tmp
for rep in 1 2; do for w in 0 4 8 11 15; do
  VS_WHATIF_OK=1 VS_WHATIF=$w timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-feat-fwd 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train whatif $w', d['value'], d['ms_per_step'])"
done; done
;;
*)
echo "usage: tools/ab.sh <experiment> [args]; experiments: r5_final r2_final r3_ab_noslp r3_ab_wgrad_unsplit r3_batch_probe r3_batch_probe_large r3_final r3_parity r4_ab_deep r4_ab_lib r4_baseline_tables r4_batch_probe r4_bnred_ab r4_check r4_deep_first r4_direct_tpw r4_final r4_halo_pmc r4_knob_sweep r4_korder_timeline r4_kstats r4_kstats32 r4_nt_ab r4_plan_ab r4_reduce_ab r4_splitk_il r4_wgrad_align_probe r4_wgrad_deep r4_wgrad_deep_step r4_wgrad_half r4_wgrad_stag r4_wgrad_whatif r4_wgrad_xcd r4_whatif"; exit 2
;;
esac
