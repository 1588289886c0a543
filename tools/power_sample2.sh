#!/bin/bash
# rocm-smi power / sclk samples while the benchmark's step runs back to back in a given arithmetic:
#   tools/power_sample2.sh <gemm> [variant.so]      (GPU box; output gpurun_out/power_<gemm>.txt)
cd "$(dirname "$0")/.."
G=${1:-fp16x2}
[ -n "$2" ] && export PURESOUND_HIP_LIB=$2
mkdir -p gpurun_out
timeout -k 10 200 python3 tools/step_time.py $G 1200 > gpurun_out/power_step_$G.log 2>&1 &
BP=$!
sleep 12
for i in $(seq 1 25); do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk" | tr '\n' ' '
  echo
  sleep 0.3
done > gpurun_out/power_raw_$G.txt
wait $BP
{ echo "# rocm-smi samples 0.3 s apart while tools/step_time.py $G 1200 runs (the benchmark's step back to back)";
  grep "ms/step" gpurun_out/power_step_$G.log;
  grep -o 'sclk clock level: [^ ]* ([0-9]*Mhz)\|Power (W): [0-9.]*' gpurun_out/power_raw_$G.txt | paste -sd' ' | sed 's/sclk clock level: [0-9]*: //g' | fold -w 160; } > gpurun_out/power_$G.txt
cat gpurun_out/power_$G.txt
