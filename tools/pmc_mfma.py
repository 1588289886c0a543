"""MFMA utilisation per kernel from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE).

usage: pmc_mfma.py PMC_DIR > profiles/<round>_pmc_mfma_busy_<tag>.txt
SQ_VALU_MFMA_BUSY_CYCLES sums the cycles every SIMD's matrix pipe was busy (32 per v_mfma_f32_32x32x16_bf16, 64 per
v_mfma_f32_32x32x2_f32); GRBM_GUI_ACTIVE sums the active cycles of the 8 XCDs (MI355X_MICROARCH.md).  Utilisation =
busy / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs); the clock the kernel ran at = GRBM_GUI_ACTIVE / 8 / duration.
"""
import csv, glob, os, re, sys
from collections import defaultdict

tot = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(int)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cnt[k] += 1
print("per-launch averages; utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)")
for k in sorted(tot, key=lambda k: -tot[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)):
    if not k.startswith("ps::") or not cnt[k]:
        continue
    busy, act = tot[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / cnt[k], tot[k]["GRBM_GUI_ACTIVE"] / cnt[k]
    print(f"{k:60s} launches {cnt[k]:5d}  MFMA busy cycles {busy:14.0f}  elapsed cycles (GRBM_GUI_ACTIVE/8) {act / 8:10.0f}  "
          f"MFMA utilisation {busy / (act / 8 * 1024):6.3f}")
