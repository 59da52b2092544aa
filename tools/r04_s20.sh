#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/s20
mkdir -p $O
(for v in base abl1 abl2 abl3 abl4 abl7 abl15 base; do
  CTCFA_ALLOW_TUNING_BUILD=1 CTCFA_LIB=$PWD/variants/$v.so timeout -k 10 200 python tools/env_sweep.py $v --steps 200 2>&1 | grep -v amdgpu.ids | cut -c1-170
done) | tee $O/ablations_final.txt
CTCFA_ALLOW_TUNING_BUILD=1 CTCFA_LIB=$PWD/variants/trace4.so timeout -k 10 200 python tools/trace4.py 2 > $O/trace4_512.txt 2>&1; grep -v amdgpu.ids $O/trace4_512.txt | head -14
