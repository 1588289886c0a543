#!/bin/bash
# Rebuild conv1x1_bf16.o with each macro setting and time the three Conv-TasNet GEMM shapes (GPU box).
# usage: tools/sweep_build.sh "<-D...>" "<-D...>" ...   (each argument = one EXTRA setting)
cd "$(dirname "$0")/.."
for extra in "$@"; do
  touch puresound_amd/csrc/conv1x1_bf16.hip
  make -C puresound_amd/csrc EXTRA="$extra" > /dev/null 2>&1 || { echo "build failed: $extra"; exit 1; }
  echo "== EXTRA=$extra"
  timeout -k 10 120 python tools/ablate_conv.py 2>/dev/null | sed -e 's/full=[0-9]*us  //' -e 's/bf16x3\/simple[^)]*)  //' -e 's/bf16x[13]\/pp[^)]*)  //g'
done
