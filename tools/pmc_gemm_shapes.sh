#!/bin/bash
root=$(pwd); cd /tmp && export TMPDIR=/tmp && cd "$root"
out=gpurun_out/pmc_gemm; rm -rf $out; mkdir -p $out
i=0
for grp in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_REQ_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $out/p$i -- python3 tools/pmc_gemm_shapes.py > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, re
from collections import defaultdict
tot, cnt = defaultdict(float), defaultdict(int)
for f in glob.glob("gpurun_out/pmc_gemm/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")
        if "il_kernel" not in k: continue
        tot[(k, row["Counter_Name"])] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
for k in sorted({k for k, _ in tot}):
    print(k)
    for (kk, c), v in sorted(tot.items()):
        if kk == k: print(f"    {c:44s} {v / cnt[(kk, c)]:16.0f}")
PY
