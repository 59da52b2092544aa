#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/s18
mkdir -p $O
echo "== pytest" && timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
echo "== round 3's library against this round's (same box; fill alone | pipelined)"
for args in "" "--vocab 38" "--vocab 64" "--vocab 29" "--vocab 48" "--segments 4096 --steps 100" "--frames 8000 --steps 100" "--utts 8 --utt-len 30" "--segments 128"; do
  echo "-- $args"
  CTCFA_ALLOW_TUNING_BUILD=1 CTCFA_LIB=$PWD/variants/r3full.so timeout -k 10 300 python tools/env_sweep.py r3 $args 2>&1 | grep -v amdgpu.ids | cut -c1-215
  timeout -k 10 300 python tools/env_sweep.py r4 $args 2>&1 | grep -v amdgpu.ids | cut -c1-215
done 2>&1 | tee $O/ab.txt
