#!/bin/bash
# kernel-trace totals of BASELINE config 4 (DPRNN, 32 x 4 s) with the masker in the fp16x2 arithmetic (GPU box): tools/prof_cfg4.sh
root=$(pwd); cd /tmp && export TMPDIR=/tmp && cd "$root"
out=gpurun_out/prof_cfg4; rm -rf $out
PS_PRECS=${PS_PRECS:-fp16x2} rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tools/preset_sweep.py cfg4_short > gpurun_out/prof_cfg4.log 2>&1
grep "^{" gpurun_out/prof_cfg4.log
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_cfg4/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
with open("gpurun_out/prof_cfg4_kernels.txt", "w") as o:
    for r in rows[:14]:
        line = (f'{r["Name"][:86]:86s} {float(r["Calls"])/5:7.1f}/fwd {float(r["AverageNs"])/1e3:9.2f} us '
                f'{float(r["TotalDurationNs"])/5/1e6:7.3f} ms/fwd {float(r["Percentage"]):5.1f}%')
        print("  ", line); o.write(line + "\n")
PY
rm -rf $out
