#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/s13
mkdir -p $O
for w in 2 1 3; do CTCFA_SB_WINDOWS=$w timeout -k 10 200 python tools/env_sweep.py "windows$w" --steps 400 2>&1 | grep -v amdgpu.ids | cut -c1-215; done | tee $O/strider_windows.txt
for m in 15 8 25; do CTCFA_SB_MARGIN_RT=$m timeout -k 10 200 python tools/env_sweep.py "base" --steps 400 2>&1 | grep -v amdgpu.ids | cut -c1-215; break; done
