#!/bin/bash
# kernel-trace totals of one preset's forward at 32 x 4 s (GPU box): tools/prof_preset.sh PRESET   (see tools/preset_sweep.py)
root=$(pwd); cd /tmp && export TMPDIR=/tmp && cd "$root"
name=$1
out=gpurun_out/prof_$name; rm -rf $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tools/preset_sweep.py $name > gpurun_out/prof_$name.log 2>&1
grep "^{" gpurun_out/prof_$name.log
python3 - "$name" <<'PY'
import csv, glob, sys
f = glob.glob(f"gpurun_out/prof_{sys.argv[1]}/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
import os
fw = 5.0 * len(os.environ.get("PS_PRECS", "fp32,fp16x2").split(","))   # per arithmetic: 2 warm-up + 3 timed forwards
with open(f"gpurun_out/prof_{sys.argv[1]}_kernels.txt", "w") as o:
    for r in rows[:22]:
        line = (f'{r["Name"][:86]:86s} {float(r["Calls"])/fw:7.1f}/fwd {float(r["AverageNs"])/1e3:9.2f} us '
                f'{float(r["TotalDurationNs"])/fw/1e6:7.3f} ms/fwd {float(r["Percentage"]):5.1f}%')
        print("  ", line); o.write(line + "\n")
PY
rm -rf $out
