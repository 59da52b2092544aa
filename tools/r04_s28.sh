#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s28
rm -f gpurun_out/s28/*.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/s28/pytest.log 2>&1 || { tail -40 gpurun_out/s28/pytest.log; exit 1; }
tail -2 gpurun_out/s28/pytest.log
for v in 29 20 5; do
timeout -k 10 200 python tools/env_sweep.py v$v --vocab $v 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/s28/sweep.log || exit 1
done
timeout -k 10 200 python tools/env_sweep.py v32 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/s28/sweep.log || exit 1
