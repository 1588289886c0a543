#!/bin/bash
# usage: tools/sweep_stamps.sh "<EXTRA>:<flags>" ...  : build a -DPS_PP_STAMPS <EXTRA> variant (tools/_variants/, the shipped
# library stays untouched) and print the s_memtime buckets of the fp16x2 GEMM
cd "$(dirname "$0")/.."
i=0
for arg in "$@"; do
  i=$((i + 1))
  extra="${arg%%:*}"; flags="${arg##*:}"
  tools/build_variant.sh stamps$i "-DPS_PP_STAMPS $extra" > /dev/null 2>&1 || { echo "build failed: $extra"; exit 1; }
  echo "== EXTRA=$extra flags=$flags"
  PURESOUND_HIP_LIB=tools/_variants/stamps$i.so timeout -k 10 120 python tools/time_f16x2.py $flags --stamps --nocheck 2>/dev/null
done
