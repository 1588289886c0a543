#!/bin/bash
# Time the three Conv-TasNet GEMM shapes for each macro setting (GPU box).  Every setting is built into a library of
# its own (tools/build_variant.sh -> tools/_variants/<name>.so, selected with PURESOUND_HIP_LIB): the shipped
# puresound_amd/libpuresound_hip.so is never rebuilt with experimental flags.
# usage: tools/sweep_build.sh "<-D...>" "<-D...>" ...   (each argument = one EXTRA setting)
cd "$(dirname "$0")/.."
i=0
for extra in "$@"; do
  i=$((i + 1))
  tools/build_variant.sh sweep$i "$extra" > /dev/null 2>&1 || { echo "build failed: $extra"; exit 1; }
  echo "== EXTRA=$extra"
  PURESOUND_HIP_LIB=tools/_variants/sweep$i.so timeout -k 10 120 python tools/time_f16x2.py 2>/dev/null
done
