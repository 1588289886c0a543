#!/bin/bash
# kernel-trace averages of BASELINE config 3 in its bf16 arithmetic (GPU box): tools/prof_cfg3_bf16.sh
root=$(pwd); cd /tmp && export TMPDIR=/tmp && cd "$root"
out=gpurun_out/prof_cfg3; rm -rf $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 -c "
import sys; sys.path.insert(0, 'tools'); import bench_configs as BC
print(BC.cfg3('cuda:0', steps=10, warmup=2, modes=('bf16',)))
" > gpurun_out/prof_cfg3.log 2>&1
grep "workload" gpurun_out/prof_cfg3.log | cut -c1-400
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_cfg3/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print(f'   {r["Name"][:100]:100s} {r["Calls"]:>6s} {float(r["AverageNs"])/1e3:8.2f} {float(r["Percentage"]):5.1f}%')
PY
rm -rf $out
