#!/bin/bash
# kernel stats + HBM-side traffic of a bench_recurrent config (GPU box): tools/pmc_any.sh dpcrn "<extra args>"
cfg=$1; extra=$2
root=$(pwd); cd /tmp && export TMPDIR=/tmp && cd "$root"
o=gpurun_out/pmc_any; rm -rf $o; mkdir -p $o
rocprofv3 --kernel-trace --stats --output-format csv -d $o/k -- python3 tools/bench_recurrent.py --which $cfg $extra > $o/line.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $o/f -- python3 tools/bench_recurrent.py --which $cfg --steps 2 --warmup 1 $extra > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $o/w -- python3 tools/bench_recurrent.py --which $cfg --steps 2 --warmup 1 $extra > /dev/null 2>&1
grep "^{" $o/line.log | cut -c1-600
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/pmc_any/k/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print(f'   {r["Name"][:84]:84s} {r["Calls"]:>6s} {float(r["AverageNs"])/1e3:9.2f} {r["Percentage"]:>6s}')
PY
python3 tools/pmc_summary.py $o/f $o/w | cut -c1-190 | head -16
rm -rf $o
