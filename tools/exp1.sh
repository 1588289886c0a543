#!/bin/bash
# round-3 experiment 1: ring depth, stamps, stagger (GPU box)
cd "$(dirname "$0")/.."
V=tools/_variants
run() { timeout -k 10 120 "$@" 2>&1 | grep -v "^$" ; }
echo "### depth variants"
for v in base v34 v25; do PURESOUND_HIP_LIB=$V/$v.so run python tools/time_f16x2.py; done
echo "### stamps"
for v in base_st v34_st; do
  for f in 0 0x1000000 0x2000000 0x3000000 0x4000000 0x5000000; do
    PURESOUND_HIP_LIB=$V/$v.so run python tools/time_f16x2.py $f --stamps --nocheck
  done
done
echo "### stagger (v34_tune)"
for g in 2 4; do for d in 4000 8000 16000; do
  echo "groups $g delay $d (stats+plain+res)"
  PS_IL_GROUPS=$g PS_IL_DELAY_STATS=$d PS_IL_DELAY_PLAIN=$d PS_IL_DELAY_RES=$d PURESOUND_HIP_LIB=$V/v34_tune.so run python tools/time_f16x2.py --nocheck
done; done
echo "groups 4 delay 0"
PS_IL_GROUPS=4 PS_IL_DELAY_STATS=0 PS_IL_DELAY_PLAIN=0 PS_IL_DELAY_RES=0 PURESOUND_HIP_LIB=$V/v34_tune.so run python tools/time_f16x2.py --nocheck
