#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/s16
mkdir -p $O
for args in "" "--frames 8000 --steps 100"; do
echo "-- $args"
for p in 0 1 2 3; do CTCFA_SB_PRIO=$p timeout -k 10 200 python tools/env_sweep.py "strider_prio$p" --steps 300 $args 2>&1 | grep -v amdgpu.ids | cut -c1-215; done
CTCFA_SB_WAVES=4 timeout -k 10 200 python tools/env_sweep.py "striders4" --steps 300 $args 2>&1 | grep -v amdgpu.ids | cut -c1-215
CTCFA_SB_WAVES=4 CTCFA_SB_PRIO=1 timeout -k 10 200 python tools/env_sweep.py "striders4p1" --steps 300 $args 2>&1 | grep -v amdgpu.ids | cut -c1-215
CTCFA_PROD_PRIO=0 timeout -k 10 200 python tools/env_sweep.py "prod0" --steps 300 $args 2>&1 | grep -v amdgpu.ids | cut -c1-215
CTCFA_PROD_PRIO=2 timeout -k 10 200 python tools/env_sweep.py "prod2" --steps 300 $args 2>&1 | grep -v amdgpu.ids | cut -c1-215
done 2>&1 | tee $O/knobs.txt
