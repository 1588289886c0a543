#!/bin/bash
# per-dispatch durations of the kernels matching PATTERN in one preset's forward (GPU box): tools/prof_launches.sh PRESET PATTERN
root=$(pwd); cd /tmp && export TMPDIR=/tmp && cd "$root"
name=$1; pat=$2
out=gpurun_out/prof_l_$name; rm -rf $out
PS_PRECS=${PS_PRECS:-fp16x2} rocprofv3 --kernel-trace --output-format csv -d $out -- python3 tools/preset_sweep.py $name > gpurun_out/prof_l_$name.log 2>&1
python3 - "$name" "$pat" <<'PY'
import csv, glob, sys
t = glob.glob(f"gpurun_out/prof_l_{sys.argv[1]}/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(t)) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = len(rows) // 5
for r in rows[-per:]:
    print(f'   {r["Kernel_Name"][:60]:60s} grid {r["Grid_Size_X"]}x{r["Grid_Size_Y"]}x{r["Grid_Size_Z"]} wg {r["Workgroup_Size_X"]} {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:9.1f} us')
PY
rm -rf $out
