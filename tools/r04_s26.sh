#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s26
rm -f gpurun_out/s26/*.log
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu > gpurun_out/s26/pytest.log 2>&1 || { tail -40 gpurun_out/s26/pytest.log; exit 1; }
tail -2 gpurun_out/s26/pytest.log
timeout -k 10 200 python tools/env_sweep.py v32 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/s26/sweep.log || exit 1
for v in 38 64 200; do
  timeout -k 10 200 python tools/env_sweep.py v${v}_narrow --vocab $v --alphabet 28 --narrow 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/s26/sweep.log || exit 1
done
timeout -k 10 200 python tools/env_sweep.py v29 --vocab 29 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/s26/sweep.log || exit 1
for v in 38; do
  echo "== vocab $v" >> gpurun_out/s26/trace.log
  CTCFA_TRACE_V=$v CTCFA_ALLOW_TUNING_BUILD=1 CTCFA_LIB=$PWD/variants/trace4.so timeout -k 10 200 python tools/trace4.py 2 2>&1 | grep -v amdgpu.ids >> gpurun_out/s26/trace.log || exit 1
done
grep "== vocab\|producer 0" gpurun_out/s26/trace.log | cut -c1-250
