#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/s8
mkdir -p $O
for v in base abl256 abl4 pace1 base; do
  CTCFA_ALLOW_TUNING_BUILD=1 CTCFA_LIB=$PWD/variants/$v.so timeout -k 10 200 python tools/env_sweep.py $v --steps 300 > $O/sweep_$v.txt 2>&1; grep -v amdgpu.ids $O/sweep_$v.txt | cut -c1-200
done
