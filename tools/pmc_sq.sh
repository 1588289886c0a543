#!/bin/bash
# SQ-side counters of the fp16x2 GEMM on the three Conv-TasNet shapes (GPU box): tools/pmc_sq.sh [variant.so]
[ -n "$1" ] && export PURESOUND_HIP_LIB=$(pwd)/$1
root=$(pwd); cd /tmp && export TMPDIR=/tmp && cd "$root"
out=gpurun_out/pmc_sq; rm -rf $out; mkdir -p $out
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u > $out/avail_sq.txt
i=0
for grp in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" \
           "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INST_CYCLES_VMEM SQ_INSTS_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM" \
           "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_F16 SQ_IFETCH"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $grp --output-format csv -d $out/p$i -- python3 tools/pmc_gemm_shapes.py > $out/p$i.log 2>&1 || echo "group $i failed: $grp"
done
python3 - <<'PY'
import csv, glob, re
from collections import defaultdict
tot, cnt = defaultdict(float), defaultdict(int)
for f in glob.glob("gpurun_out/pmc_sq/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")
        if "il_kernel" not in k and "w1_kernel" not in k and "rb_kernel" not in k: continue
        tot[(k, row["Counter_Name"])] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
for k in sorted({k for k, _ in tot}):
    print(k)
    for (kk, c), v in sorted(tot.items()):
        if kk == k: print(f"    {c:44s} {v / cnt[(kk, c)]:16.0f}")
PY
