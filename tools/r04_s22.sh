#!/bin/bash
# narrowed plans: parity, then vocab 38 with / without labels, vocab 64/100
set -o pipefail
mkdir -p gpurun_out/s22
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "narrowed or vocabulary" > gpurun_out/s22/pytest.log 2>&1 || { tail -40 gpurun_out/s22/pytest.log; exit 1; }
tail -2 gpurun_out/s22/pytest.log
for v in 38 64 100; do
  timeout -k 10 200 python tools/env_sweep.py v${v}_plain --vocab $v --alphabet 28 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/s22/sweep.log || exit 1
  timeout -k 10 200 python tools/env_sweep.py v${v}_narrow --vocab $v --alphabet 28 --with-labels 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/s22/sweep.log || exit 1
done
timeout -k 10 200 python tools/env_sweep.py v32 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/s22/sweep.log || exit 1
timeout -k 10 200 python tools/env_sweep.py v32_1set --sets 1 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/s22/sweep.log
