#!/bin/bash
# kernel-trace totals of one ns_dpcrn_v0_causal forward at 32 x 4 s (GPU box): tools/prof_dpcrn.sh [fp16x2|fp32] [tag]
root=$(pwd); cd /tmp && export TMPDIR=/tmp && cd "$root"
gemm=${1:-fp16x2}; tag=${2:-dpcrn}
out=gpurun_out/prof_$tag; rm -rf $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tools/bench_recurrent.py --which dpcrn --gemm $gemm --steps 5 > gpurun_out/prof_$tag.log 2>&1
grep ms_per_forward gpurun_out/prof_$tag.log | cut -c1-300
python3 - "$tag" <<'PY'
import csv, glob, sys
f = glob.glob(f"gpurun_out/prof_{sys.argv[1]}/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
fw = 8.0   # 3 warm-up + 5 timed forwards in the profiled process
with open(f"gpurun_out/prof_{sys.argv[1]}_kernels.txt", "w") as o:
    for r in rows[:24]:
        line = (f'{r["Name"][:86]:86s} {float(r["Calls"])/fw:7.1f}/fwd {float(r["AverageNs"])/1e3:9.2f} us '
                f'{float(r["TotalDurationNs"])/fw/1e6:7.3f} ms/fwd {float(r["Percentage"]):5.1f}%')
        print("  ", line); o.write(line + "\n")
# the ten convolutions of the last forward, in launch order (5 down, 5 up)
t = glob.glob(f"gpurun_out/prof_{sys.argv[1]}/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(t)) if "conv2d_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
with open(f"gpurun_out/prof_{sys.argv[1]}_kernels.txt", "a") as o:
    for r in rows[-10:]:
        line = f'conv2d launch: {r["Kernel_Name"][:40]:40s} grid {r["Grid_Size_X"]}x{r["Grid_Size_Y"]}x{r["Grid_Size_Z"]} {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:9.1f} us'
        print("  ", line); o.write(line + "\n")
PY
rm -rf $out
