#!/bin/bash
cd "$(dirname "$0")/.."
export PURESOUND_HIP_LIB=tools/_variants/v34.so
for cfg in "1 0" "2 0" "2 128" "2 192" "4 64" "4 128" "2 160"; do
  set -- $cfg
  PS_STREAMS=$1 PS_CAP=$2 timeout -k 10 200 python tools/step_time.py fp16x2 20 2>&1 | grep "ms/step"
done
