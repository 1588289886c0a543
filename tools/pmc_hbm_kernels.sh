#!/bin/bash
# Hardware counters of the HBM-bound kernels (dwconv, encoder, decoder) at 32 utterances, one group per pass.
root=$(pwd); cd /tmp && export TMPDIR=/tmp && cd "$root"
out=gpurun_out/pmc_hbmk; rm -rf $out; mkdir -p $out
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR" \
           "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_STALL" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_REQ_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $out/p$i -- python3 tools/bench_hbm_kernels.py > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, re
from collections import defaultdict
tot, cnt = defaultdict(float), defaultdict(int)
for f in glob.glob("gpurun_out/pmc_hbmk/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")
        if not k.startswith("ps::"): continue
        tot[(k, row["Counter_Name"])] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
ks = sorted({k for k, _ in tot})
for k in ks:
    print(k)
    for (kk, c), v in sorted(tot.items()):
        if kk == k: print(f"    {c:44s} {v / cnt[(kk, c)]:16.0f}")
PY
