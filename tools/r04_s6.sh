#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/s6
mkdir -p $O
echo "== producer ablations: 16 no LDS stores, 48 + no vector work, 64 b64 stores, 128 no start-column store; 7 = no waits/hand-over/tile stores, 23 = 7+16, 55 = 7+48, 15 = 7 + no producer"
for v in base abl16 abl48 abl64 abl128 abl7 abl23 abl55 abl15 base; do
  CTCFA_ALLOW_TUNING_BUILD=1 CTCFA_LIB=$PWD/variants/$v.so timeout -k 10 200 python tools/env_sweep.py $v --steps 200 > $O/sweep_$v.txt 2>&1; grep -v amdgpu.ids $O/sweep_$v.txt | cut -c1-120
done
