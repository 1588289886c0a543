#!/bin/bash
# kernel traces of configs 4 and 5 (GPU box): tools/prof_recurrent.sh TAG -> gpurun_out/prof_<TAG>_cfg{4,5}_kernel_stats.csv
tag=${1:-r03}
root=$(pwd); cd /tmp && export TMPDIR=/tmp && cd "$root"
for c in cfg4 cfg5; do
  out=gpurun_out/prof_${tag}_$c; rm -rf $out
  extra="--gemm fp16x2"; [ $c = cfg5 ] && extra="--chunks 300"
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tools/bench_recurrent.py --which $c $extra > gpurun_out/prof_${tag}_$c.log 2>&1
  cp $(ls $out/*/*kernel_stats.csv | head -1) gpurun_out/prof_${tag}_${c}_kernel_stats.csv
  grep '^{' gpurun_out/prof_${tag}_$c.log > gpurun_out/prof_${tag}_${c}_line.json
  rm -rf $out
done
