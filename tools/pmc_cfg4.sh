#!/bin/bash
# HBM-side traffic of config 4's kernels (GPU box): FETCH_SIZE and WRITE_SIZE in separate passes
root=$(pwd); cd /tmp && export TMPDIR=/tmp && cd "$root"
o=gpurun_out/pmc_cfg4; rm -rf $o; mkdir -p $o
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $o/f -- python3 tools/bench_recurrent.py --which cfg4 --gemm fp16x2 --steps 2 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $o/w -- python3 tools/bench_recurrent.py --which cfg4 --gemm fp16x2 --steps 2 --warmup 1 > /dev/null 2>&1
python3 tools/pmc_summary.py $o/f $o/w > gpurun_out/pmc_cfg4_traffic.txt
rm -rf $o
cat gpurun_out/pmc_cfg4_traffic.txt | cut -c1-200
