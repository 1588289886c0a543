#!/bin/bash
# usage: tools/prof_recurrent.sh TAG WHICH  -> rocprofv3 kernel trace of tools/bench_recurrent.py (cfg4 or cfg5)
TAG=$1; WHICH=$2; FLAGS=${3:-0}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 tools/bench_recurrent.py --which $WHICH --chunks 60 --steps 5 --flags $FLAGS > gpurun_out/benchrec_$TAG.log 2>&1
tail -3 gpurun_out/benchrec_$TAG.log
f=$(ls gpurun_out/prof_$TAG/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:14]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>7s} avg_us={float(r['AverageNs'])/1e3:9.2f} tot_ms={float(r['TotalDurationNs'])/1e6:9.2f} {r['Percentage']}%")
PY
