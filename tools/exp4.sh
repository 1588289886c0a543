#!/bin/bash
cd "$(dirname "$0")/.."
timeout -k 10 300 python tools/time_f16x2.py 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python tools/time_f16x2.py 128 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python tools/step_time.py fp16x2 20 2>&1 | grep "ms/step"
