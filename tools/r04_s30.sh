#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s30
rm -rf gpurun_out/s30/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/s30/prof -- python3 tools/windowed_one.py 9500 40 30 5 > gpurun_out/s30/run.log 2>&1 || { tail -20 gpurun_out/s30/run.log; exit 1; }
grep "call" gpurun_out/s30/run.log
find gpurun_out/s30/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cut -c1-200 {} | head -8'
