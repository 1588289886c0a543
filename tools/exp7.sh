#!/bin/bash
cd "$(dirname "$0")/.."
export PURESOUND_HIP_LIB=tools/_variants/tune.so
for cfg in "4 16000" "2 16000" "2 24000" "2 12000" "3 16000" "2 32000" "4 16000" "2 16000" "8 8000" "1 0"; do
  set -- $cfg
  echo -n "groups $1 delay $2: "
  PS_IL_GROUPS=$1 PS_IL_DELAY_RES=$2 timeout -k 10 200 python tools/step_time.py fp16x2 20 2>&1 | grep "ms/step" | cut -c40-140
done
