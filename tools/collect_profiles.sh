#!/bin/bash
# Collect the round's judged profiles on the GPU box (run through gpurun from the repo root):
#   kernel trace + stats of the default bench.py command, PMC HBM-traffic passes (FETCH_SIZE / WRITE_SIZE separately)
# usage: tools/collect_profiles.sh r03
set -e
tag=${1:-r04}
root=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd "$root"
out=gpurun_out/prof_$tag
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $out/bench_under_rocprof.log 2>&1
grep '^{' $out/bench_under_rocprof.log > $out/bench_line_under_rocprof.json
for gemm in fp16x2 bf16x3 fp32; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_f_$gemm -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --gemm $gemm > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_w_$gemm -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --gemm $gemm > /dev/null 2>&1
  python3 tools/pmc_summary.py $out/pmc_f_$gemm $out/pmc_w_$gemm $out/pmc_hbm_traffic_$gemm.json > $out/pmc_hbm_traffic_$gemm.txt
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_mfma -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python3 tools/pmc_mfma.py $out/pmc_mfma > $out/pmc_mfma_busy.txt 2>&1 || true
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_mfma3 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --gemm bf16x3 > /dev/null 2>&1
(echo; echo "# the same with --gemm bf16x3"; python3 tools/pmc_mfma.py $out/pmc_mfma3) >> $out/pmc_mfma_busy.txt 2>&1 || true
cp $(ls $out/kt/*/*kernel_stats.csv | head -1) $out/bench_kernel_stats.csv
rm -rf $out/pmc_f_* $out/pmc_w_* $out/pmc_mfma $out/pmc_mfma3 $out/kt
python3 bench.py --steps 20 --warmup 5 2>/dev/null | grep "^{" > $out/bench_line.json
ls -la $out
