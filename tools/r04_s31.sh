#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s31
rm -rf gpurun_out/s31/*
for d in 0; do
CTCFA_WIN_DEBUG=$d timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/s31/prof$d -- python3 tools/windowed_one.py 9500 40 30 4 > gpurun_out/s31/run$d.log 2>&1 || { tail -20 gpurun_out/s31/run$d.log; exit 1; }
echo "debug $d"; find gpurun_out/s31/prof$d -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'grep -E "band|windowed" {} | cut -d, -f2-4 | sed "s/.*)\",//"'
find gpurun_out/s31/prof$d -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'grep -E "band|windowed" {} | awk -F"\"," "{print \$2,\$3,\$4}" | cut -c1-80'
done
