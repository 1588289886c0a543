#!/bin/bash
cd "$(dirname "$0")/.."
for i in 1 2; do
for v in "" prio0 nq1 prio2; do
  lib=""; [ -n "$v" ] && lib=tools/_variants/$v.so
  echo -n "${v:-default}: "
  PURESOUND_HIP_LIB=$lib timeout -k 10 200 python tools/step_time.py fp16x2 20 2>&1 | grep ms/step | cut -c40-130
done; done
