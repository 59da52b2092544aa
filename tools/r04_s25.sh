#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s25
for v in 32 38 64; do
  echo "== vocab $v" | tee -a gpurun_out/s25/trace.log
  CTCFA_TRACE_V=$v CTCFA_ALLOW_TUNING_BUILD=1 CTCFA_LIB=$PWD/variants/trace4.so timeout -k 10 200 python tools/trace4.py 2 2>&1 | grep -v amdgpu.ids >> gpurun_out/s25/trace.log || exit 1
done
tail -5 gpurun_out/s25/trace.log
