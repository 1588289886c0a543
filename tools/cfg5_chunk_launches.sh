#!/bin/bash
# launches per steady-state chunk of config 5, by kernel name: the difference of two traces (300 and 600 chunks) / 300
root=$(pwd); cd /tmp && export TMPDIR=/tmp && cd "$root"
for n in 300 600; do
  out=gpurun_out/c5_$n; rm -rf $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tools/bench_recurrent.py --which cfg5 --chunks $n > /dev/null 2>&1
  cp $(ls $out/*/*kernel_stats.csv | head -1) gpurun_out/c5_$n.csv; rm -rf $out
done
python3 - <<'PY'
import csv
a = {r["Name"]: int(r["Calls"]) for r in csv.DictReader(open("gpurun_out/c5_300.csv"))}
b = {r["Name"]: int(r["Calls"]) for r in csv.DictReader(open("gpurun_out/c5_600.csv"))}
print("launches per steady-state chunk (600-chunk trace minus 300-chunk trace, / 300):")
for k in sorted(b, key=lambda k: -(b[k] - a.get(k, 0))):
    d = (b[k] - a.get(k, 0)) / 300.0
    if d > 0.004:
        print(f"  {d:8.2f}  {k[:110]}")
PY
