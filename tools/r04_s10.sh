#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/s10
mkdir -p $O
echo "== which of this round's changes to the tile loop pay, alone and beside the backtrack (same box): config 3, vocab 38"
for args in "" "--vocab 38"; do
  echo "-- $args"
  for v in r3full f_old3 f_bb1 f_mask f_nodefer f_mask_nodefer PRODUCT r3full; do
    if [ $v = PRODUCT ]; then timeout -k 10 300 python tools/env_sweep.py r4 $args 2>&1 | grep -v amdgpu.ids | cut -c1-215
    else CTCFA_ALLOW_TUNING_BUILD=1 CTCFA_LIB=$PWD/variants/$v.so timeout -k 10 300 python tools/env_sweep.py $v $args 2>&1 | grep -v amdgpu.ids | cut -c1-215; fi
  done
done
