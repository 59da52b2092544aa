#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/s21
mkdir -p $O
(for args in "--vocab 38" "--vocab 48" "--vocab 64" "--vocab 56"; do
  echo "-- $args"
  timeout -k 10 300 python tools/env_sweep.py shipped $args 2>&1 | grep -v amdgpu.ids | cut -c1-215
  CTCFA_ALLOW_TUNING_BUILD=1 CTCFA_LIB=$PWD/variants/f_leanall.so timeout -k 10 300 python tools/env_sweep.py leanall $args 2>&1 | grep -v amdgpu.ids | cut -c1-215
done) | tee $O/leanall.txt
