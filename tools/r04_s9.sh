#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/s9
mkdir -p $O
echo "== round 3's kernels against this round's, same box, same process order (config 3, then vocab 38 / 64 / 29, then 4096 segments)"
for args in "" "--vocab 38" "--vocab 64" "--vocab 29" "--segments 4096 --steps 100" "--frames 8000 --steps 100" ; do
  echo "-- $args"
  CTCFA_ALLOW_TUNING_BUILD=1 CTCFA_LIB=$PWD/variants/r3full.so timeout -k 10 300 python tools/env_sweep.py r3 $args 2>&1 | grep -v amdgpu.ids | cut -c1-215
  timeout -k 10 300 python tools/env_sweep.py r4 $args 2>&1 | grep -v amdgpu.ids | cut -c1-215
done
