#!/bin/bash
# kernel-trace stats of an arbitrary python tool (GPU box): tools/prof_py.sh OUTNAME tools/x.py [args]
name=$1; shift
root=$(pwd); cd /tmp && export TMPDIR=/tmp && cd "$root"
out=gpurun_out/prof_py; rm -rf $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 "$@" > gpurun_out/$name.log 2>&1
cp $(ls $out/*/*kernel_stats.csv | head -1) gpurun_out/${name}_kernel_stats.csv
rm -rf $out
python3 - "$name" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(f"gpurun_out/{sys.argv[1]}_kernel_stats.csv")))
for r in rows[:25]:
    print(f'{r["Name"][:95]:95s} {r["Calls"]:>6s} {float(r["AverageNs"])/1e3:9.2f} {r["Percentage"]:>6s}')
PY
