#!/bin/bash
cd "$(dirname "$0")/.."
V=tools/_variants
for v in "$@"; do PURESOUND_HIP_LIB=$V/$v.so timeout -k 10 200 python tools/step_time.py 2>&1 | grep "ms/step"; done
