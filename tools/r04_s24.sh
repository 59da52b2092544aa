#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s24
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "narrowed or vocabulary or blank_transition" > gpurun_out/s24/pytest.log 2>&1 || { tail -40 gpurun_out/s24/pytest.log; exit 1; }
tail -2 gpurun_out/s24/pytest.log
for v in 38 64 200; do
  timeout -k 10 200 python tools/env_sweep.py v${v}_narrow --vocab $v --alphabet 28 --with-labels 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/s24/sweep.log || exit 1
done
timeout -k 10 200 python tools/env_sweep.py v32_1set --sets 1 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/s24/sweep.log
