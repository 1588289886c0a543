#!/bin/bash
cd "$(dirname "$0")/.."
for v in "" et1 et2 et8 et16 uc4 uc16; do
  lib=""; [ -n "$v" ] && lib=tools/_variants/$v.so
  echo -n "${v:-default}: "
  PURESOUND_HIP_LIB=$lib timeout -k 10 200 python tools/step_time.py fp16x2 20 2>&1 | grep ms/step | sed 's/.*dwconv[^|]*| //'
done
