#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s27
rm -f gpurun_out/s27/*.log
timeout -k 10 200 python tools/env_sweep.py v32_a28 --alphabet 28 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/s27/sweep.log || exit 1
timeout -k 10 200 python tools/env_sweep.py v32_a20 --alphabet 20 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/s27/sweep.log || exit 1
timeout -k 10 200 python tools/env_sweep.py v38n_a20 --vocab 38 --alphabet 20 --narrow 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/s27/sweep.log || exit 1
timeout -k 10 200 python tools/env_sweep.py v38n_a31 --vocab 38 --alphabet 31 --narrow 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/s27/sweep.log || exit 1
