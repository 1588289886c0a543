#!/bin/bash
# usage: tools_gpu_cycle.sh TAG   -> runs gpu tests, then bench under rocprofv3 kernel-trace; outputs under gpurun_out/
TAG=$1
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/test_$TAG.log 2>&1
rc=$?; tail -3 gpurun_out/test_$TAG.log
if [ $rc -ne 0 ]; then echo "TESTS FAILED rc=$rc"; exit $rc; fi
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_$TAG.log 2>&1
grep metric gpurun_out/bench_$TAG.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms/step',d['ms_per_step'],'value',d['value']); print(d.get('roofline'))"
