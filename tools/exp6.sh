#!/bin/bash
cd "$(dirname "$0")/.."
for i in 1 2; do
for v in "" tools/_variants/noslp.so; do
  for f in 0 128; do
    PURESOUND_HIP_LIB=$v PS_FLAGS=$f timeout -k 10 200 python tools/step_time.py fp16x2 20 2>&1 | grep "ms/step" | cut -c1-120
  done
done
done
