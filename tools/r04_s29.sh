#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s29
rm -rf gpurun_out/s29/*
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "window or spans or label_matri" > gpurun_out/s29/pytest.log 2>&1 || { tail -60 gpurun_out/s29/pytest.log | cut -c1-300; exit 1; }
tail -2 gpurun_out/s29/pytest.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/s29/prof -- python3 tools/windowed_one.py 9500 40 30 5 > gpurun_out/s29/run.log 2>&1 || { tail -20 gpurun_out/s29/run.log; exit 1; }
grep "call" gpurun_out/s29/run.log
find gpurun_out/s29/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cut -c1-120 {} | head -4'
timeout -k 10 600 python tools/windowed_timing.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s29/timing.log
