#!/bin/bash
# samples rocm-smi power / clocks while bench.py runs (is the step power-limited?)
mkdir -p gpurun_out
timeout -k 10 300 python3 bench.py --steps 400 --warmup 5 > gpurun_out/power_bench.log 2>&1 &
BP=$!
sleep 8
for i in $(seq 1 90); do
  rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "Power|sclk|GPU use|mclk" | tr '\n' ' '
  echo
  sleep 0.4
done > gpurun_out/power_samples.txt
wait $BP
tail -c 600 gpurun_out/power_bench.log | head -c 300
echo
grep -o 'sclk clock level: [^ ]* ([0-9]*Mhz)\|Power (W): [0-9.]*' gpurun_out/power_samples.txt | paste -sd' ' | fold -w 200
