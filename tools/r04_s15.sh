#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/s15
mkdir -p $O
echo "== pytest" && timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
echo "== one row per LDS store, two producers, explicit load waits (r4) against the 16-byte producer (f_noaddtid) and round 3's library"
for args in "" "--segments 4096 --steps 100" "--frames 8000 --steps 100" "--utts 8 --utt-len 30" "--segments 128"; do
  echo "-- $args"
  CTCFA_ALLOW_TUNING_BUILD=1 CTCFA_LIB=$PWD/variants/r3full.so timeout -k 10 300 python tools/env_sweep.py r3 $args 2>&1 | grep -v amdgpu.ids | cut -c1-215
  CTCFA_ALLOW_TUNING_BUILD=1 CTCFA_LIB=$PWD/variants/f_noaddtid.so timeout -k 10 300 python tools/env_sweep.py noaddtid $args 2>&1 | grep -v amdgpu.ids | cut -c1-215
  timeout -k 10 300 python tools/env_sweep.py r4 $args 2>&1 | grep -v amdgpu.ids | cut -c1-215
done 2>&1 | tee $O/addtid.txt
