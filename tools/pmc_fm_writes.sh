#!/bin/bash
# WRITE_SIZE of every frame-major GEMM launch of one ns_dpcrn forward (GPU box)
root=$(pwd); cd /tmp && export TMPDIR=/tmp && cd "$root"
out=gpurun_out/pmc_fmw; rm -rf $out; mkdir -p $out
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/w -- python3 tools/bench_recurrent.py --which dpcrn --gemm fp16x2 --steps 1 --warmup 1 > $out/w.log 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/pmc_fmw/w/**/*counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "rb_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "WRITE_SIZE"]
    print(list(rows[0].keys()) if rows else "no rows")
    for r in rows[:16]:
        print(r["Kernel_Name"][30:90], r.get("Grid_Size"), r.get("Workgroup_Size"), float(r["Counter_Value"]) * 1024 / 1e6, "MB")
PY
rm -rf $out
