#!/bin/bash
# usage: tools/sweep_stamps.sh "<EXTRA>:<flags>" ...  : rebuild with -DPS_PP_STAMPS <EXTRA> and print the stamp buckets
cd "$(dirname "$0")/.."
for arg in "$@"; do
  extra="${arg%%:*}"; flags="${arg##*:}"
  touch puresound_amd/csrc/conv1x1_bf16.hip
  make -C puresound_amd/csrc EXTRA="-DPS_PP_STAMPS $extra" > /dev/null 2>&1 || { echo "build failed: $extra"; exit 1; }
  echo "== EXTRA=$extra flags=$flags"
  timeout -k 10 120 python tools/stamp_il.py $flags 2>/dev/null
done
