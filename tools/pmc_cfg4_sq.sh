#!/bin/bash
# SQ-side counters of config 4's kernels (GPU box): tools/pmc_cfg4_sq.sh -> gpurun_out/pmc_cfg4_sq.txt
root=$(pwd); cd /tmp && export TMPDIR=/tmp && cd "$root"
out=gpurun_out/pmc_cfg4_sq; rm -rf $out; mkdir -p $out
i=0
for grp in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" \
           "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_INSTS_LDS SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_WAVES"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $grp --output-format csv -d $out/p$i -- python3 tools/bench_recurrent.py --which cfg4 --gemm fp16x2 --steps 2 --warmup 1 > $out/p$i.log 2>&1 || echo "group $i failed: $grp"
done
python3 - <<'PY' > gpurun_out/pmc_cfg4_sq.txt
import csv, glob, re
from collections import defaultdict
tot, cnt = defaultdict(float), defaultdict(int)
for f in glob.glob("gpurun_out/pmc_cfg4_sq/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")
        if not k.startswith("ps::"): continue
        tot[(k, row["Counter_Name"])] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
for k in sorted({k for k, _ in tot}):
    print(k)
    for (kk, c), v in sorted(tot.items()):
        if kk == k: print(f"    {c:44s} {v / cnt[(kk, c)]:16.0f}")
PY
rm -rf $out
cat gpurun_out/pmc_cfg4_sq.txt
