#!/bin/bash
root=$(pwd); cd /tmp && export TMPDIR=/tmp && cd "$root"
out=gpurun_out/pmc_out; rm -rf $out; mkdir -p $out
for v in bf16x3 fp16x2_amax; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/f_$v -- python3 tools/pmc_out_conv.py $v > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/w_$v -- python3 tools/pmc_out_conv.py $v > /dev/null 2>&1
  echo "== $v"; python3 tools/pmc_summary.py $out/f_$v $out/w_$v | grep il_kernel | cut -c1-200
done
