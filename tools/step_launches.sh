#!/bin/bash
# launches per steady-state step of bench.py's default configuration, by kernel name: (40-step trace - 20-step trace) / 20
root=$(pwd); cd /tmp && export TMPDIR=/tmp && cd "$root"
for n in 20 40; do
  out=gpurun_out/sl_$n; rm -rf $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --steps $n --warmup 5 --no-cpu-baseline --no-other-configs --no-roofline > /dev/null 2>&1
  cp $(ls $out/*/*kernel_stats.csv | head -1) gpurun_out/sl_$n.csv; rm -rf $out
done
python3 - <<'PY'
import csv
a = {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open("gpurun_out/sl_20.csv"))}
b = {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open("gpurun_out/sl_40.csv"))}
print("launches and microseconds per steady-state step (40-step trace minus 20-step trace, / 20):")
tot = 0.0
for k in sorted(b, key=lambda k: -(b[k][1] - a.get(k, (0, 0))[1])):
    dc = (b[k][0] - a.get(k, (0, 0))[0]) / 20.0
    dt = (b[k][1] - a.get(k, (0, 0))[1]) / 20.0 / 1e3
    if dc > 0.04:
        tot += dt
        print(f"  {dc:7.2f} x  {dt:9.1f} us  {k[:120]}")
print(f"  total {tot / 1e3:.3f} ms of kernel time per step")
PY
