#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/windowed
rm -rf gpurun_out/windowed/*
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/windowed/pytest.log 2>&1 || { tail -60 gpurun_out/windowed/pytest.log | cut -c1-300; exit 1; }
tail -2 gpurun_out/windowed/pytest.log
out=gpurun_out/windowed/r04_windowed_rows.txt
echo "## tools/windowed_timing.py: the windowed regime with band_fill_kernel + the 64-step walk (this round's library)" > $out
timeout -k 10 600 python tools/windowed_timing.py 2>&1 | grep -v amdgpu.ids >> $out || exit 1
echo "## the same with CTCFA_NO_BAND_FILL=1 CTCFA_NO_FAST_WALK=1 (the literal kernel alone: rounds 1-3)" >> $out
CTCFA_NO_BAND_FILL=1 CTCFA_NO_FAST_WALK=1 timeout -k 10 900 python tools/windowed_timing.py 2>&1 | grep -v amdgpu.ids >> $out || exit 1
echo "## rocprofv3 --kernel-trace --stats: tools/windowed_one.py 9500 40 30 5 (one 190 s window, five calls)" >> $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/windowed/prof -- python3 tools/windowed_one.py 9500 40 30 5 > gpurun_out/windowed/run.log 2>&1 || { tail -20 gpurun_out/windowed/run.log; exit 1; }
grep call gpurun_out/windowed/run.log >> $out
find gpurun_out/windowed/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -4 {}' >> $out
cat $out | cut -c1-220
