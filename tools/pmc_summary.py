"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; collected separately).

usage: pmc_summary.py FETCH_DIR WRITE_DIR [OUT.json] > profiles/<round>_pmc_hbm_traffic_<tag>.txt
FETCH_SIZE / WRITE_SIZE are reported in KiB summed over the XCDs; FETCH_SIZE is doubled as MI355X_MICROARCH.md
prescribes for gfx950 (a wide coalesced read is counted at half its size).
"""
import csv, glob, os, re, sys
from collections import defaultdict


def load(d, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")
            tot[k] += float(row["Counter_Value"])
            cnt[k] += 1
    return tot, cnt


ft, fc = load(sys.argv[1], "FETCH_SIZE")
wt, wc = load(sys.argv[2], "WRITE_SIZE")
as_json = {}
print("per-launch averages; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced read)")
for k in sorted(ft, key=lambda k: -ft[k]):
    if not k.startswith("ps::"):
        continue
    n = fc[k]
    fetch = ft[k] / n * 1024 * 2 / 1e6
    write = wt.get(k, 0.0) / max(wc.get(k, 1), 1) * 1024 / 1e6
    as_json[k] = {"launches": n, "fetch_bytes": fetch * 1e6, "write_bytes": write * 1e6, "total_bytes": (fetch + write) * 1e6}
    print(f"{k:60s} launches {n:5d}  FETCH_SIZE(raw KiB) {ft[k] / n:12.0f} -> x2 corrected {fetch:9.1f} MB   "
          f"WRITE_SIZE {write:9.1f} MB   total {fetch + write:9.1f} MB")

if len(sys.argv) > 3:  # machine-readable copy (bench.py reads roofline.traffic from it)
    import json
    json.dump({"recipe": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; FETCH_SIZE x 2 (gfx950)",
               "kernels": as_json}, open(sys.argv[3], "w"), indent=1, sort_keys=True)
