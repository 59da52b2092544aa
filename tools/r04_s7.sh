#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/s7
mkdir -p $O
echo "== producer pacing (s_sleep between its LDS stores; all with the start-column entry written once)"
for v in base pace0 pace2 pace4 pace6 pace8 base; do
  CTCFA_ALLOW_TUNING_BUILD=1 CTCFA_LIB=$PWD/variants/$v.so timeout -k 10 200 python tools/env_sweep.py $v --steps 300 > $O/sweep_$v.txt 2>&1; grep -v amdgpu.ids $O/sweep_$v.txt | cut -c1-200
done
echo "== pytest (product = pace 4)" && timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
