#!/bin/bash
# HBM traffic per kernel of one ns_dpcrn_v0_causal forward (fp16x2 arithmetic), separate FETCH_SIZE / WRITE_SIZE passes
# (GPU box): tools/pmc_dpcrn.sh [tag]
root=$(pwd); cd /tmp && export TMPDIR=/tmp && cd "$root"
tag=${1:-dpcrn}
out=gpurun_out/pmc_$tag; rm -rf $out; mkdir -p $out
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/f -- python3 tools/bench_recurrent.py --which dpcrn --gemm fp16x2 --steps 1 --warmup 1 > $out/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/w -- python3 tools/bench_recurrent.py --which dpcrn --gemm fp16x2 --steps 1 --warmup 1 > $out/w.log 2>&1
python3 tools/pmc_summary.py $out/f $out/w > gpurun_out/pmc_${tag}_traffic.txt
rm -rf $out
cat gpurun_out/pmc_${tag}_traffic.txt | cut -c1-200
