#!/bin/bash
# GPU box: the shader clock and the package power while the C2 step runs back to back (lanes, then a caller's stream), sampled with
# rocm-smi twice a second: is the steady state power-limited?     bash tools/clock_probe.sh
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rocm-smi --showclocks --showpower --showmaxpower 2>/dev/null | grep -E "sclk|mclk|Power|power" | head -8
python3 bench.py --steps 60000 --warmup 20 --no-cpu-baseline --no-robustness --no-host-path > gpurun_out/clock_probe_bench.json 2>&1 &
PID=$!
sleep 6   # import + settle
for i in $(seq 1 10); do
  rocm-smi --showclocks --showpower --showtemp --showuse 2>/dev/null | grep -E "sclk|Power \(W\)|junction|GPU use" | sed 's/^GPU\[0\][[:space:]]*: //' | tr '\n' '|'; echo
  sleep 0.5
done
wait $PID
tail -c 600 gpurun_out/clock_probe_bench.json
