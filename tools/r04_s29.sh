#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s29
rm -f gpurun_out/s29/*.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "window or spans or label_matri" > gpurun_out/s29/pytest.log 2>&1 || { tail -60 gpurun_out/s29/pytest.log | cut -c1-300; exit 1; }
tail -2 gpurun_out/s29/pytest.log
timeout -k 10 600 python tools/windowed_timing.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s29/timing.log
