#!/bin/bash
# kernel-trace averages of one bench_recurrent config for a list of library variants (GPU box):
#   tools/prof_one.sh cfg4 "--gemm fp16x2" "" tools/_variants/x.so ...
cfg=$1; extra=$2; shift 2
root=$(pwd); cd /tmp && export TMPDIR=/tmp && cd "$root"
for v in "$@"; do
  out=gpurun_out/prof_one; rm -rf $out
  export PURESOUND_HIP_LIB=$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tools/bench_recurrent.py --which $cfg $extra > /dev/null 2>&1
  echo "== ${v:-shipped}"
  python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_one/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:6]:
    print(f'   {r["Name"][:70]:70s} {r["Calls"]:>6s} {float(r["AverageNs"])/1e3:8.2f}')
PY
  rm -rf $out
done
