#!/bin/bash
# what the GPU clocks / power do while the train step replays (read-only rocm-smi samples beside a long bench run)
mkdir -p gpurun_out/clock
python bench.py --steps 3000 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/clock/bench.json 2> gpurun_out/clock/bench.err &
BP=$!
sleep 25
for i in 1 2 3 4 5 6; do
  /opt/rocm/bin/rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (edge|junction|memory)" | tr -s ' ' | head -12
  echo "--"
  sleep 2
done
wait $BP
tail -c 300 gpurun_out/clock/bench.json | head -c 300; echo
echo "idle:"; sleep 3
/opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | tr -s ' ' | head -6
