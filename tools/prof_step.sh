#!/bin/bash
# kernel-trace averages of the benchmark's step (tools/step_time.py) for a list of ps_debug_flags values (GPU box):
#   tools/prof_step.sh 0 0x400000
root=$(pwd); cd /tmp && export TMPDIR=/tmp && cd "$root"
for f in "$@"; do
  out=gpurun_out/prof_step; rm -rf $out
  PS_FLAGS=$f rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tools/step_time.py fp16x2 20 > gpurun_out/prof_step_$f.log 2>&1
  echo "== flags $f: $(grep ms/step gpurun_out/prof_step_$f.log | cut -c1-80)"
  python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_step/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(f'   {r["Name"][:90]:90s} {r["Calls"]:>6s} {float(r["AverageNs"])/1e3:8.2f}')
PY
  rm -rf $out
done
