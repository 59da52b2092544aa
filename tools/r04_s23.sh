#!/bin/bash
# narrowed plans with the narrowed strider: the whole GPU suite, then vocab 38 / 64 / 100 / 200 with and without labels
set -o pipefail
mkdir -p gpurun_out/s23
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/s23/pytest.log 2>&1 || { tail -40 gpurun_out/s23/pytest.log; exit 1; }
tail -2 gpurun_out/s23/pytest.log
for v in 38 64 100 200; do
  timeout -k 10 200 python tools/env_sweep.py v${v}_plain --vocab $v --alphabet 28 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/s23/sweep.log || exit 1
  timeout -k 10 200 python tools/env_sweep.py v${v}_narrow --vocab $v --alphabet 28 --narrow 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/s23/sweep.log || exit 1
done
timeout -k 10 200 python tools/env_sweep.py v32_1set 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/s23/sweep.log
