#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/s5
mkdir -p $O
echo "== ablations (same box): base, 1 no hand-over, 2 no producer/tile waits, 3 both, 7 + no stores, 15 + no producer work"
for v in base abl1 abl2 abl3 abl7 abl15 base; do
  CTCFA_ALLOW_TUNING_BUILD=1 CTCFA_LIB=$PWD/variants/$v.so timeout -k 10 200 python tools/env_sweep.py $v --steps 200 > $O/sweep_$v.txt 2>&1; grep -v amdgpu.ids $O/sweep_$v.txt
done
echo "== other tile widths at 512 segments (product build)"
timeout -k 10 300 python tools/env_sweep.py k1 --cols-per-lane 1 --steps 200 > $O/sweep_k1.txt 2>&1; grep -v amdgpu.ids $O/sweep_k1.txt
timeout -k 10 300 python tools/env_sweep.py k3 --cols-per-lane 3 --steps 200 > $O/sweep_k3.txt 2>&1; grep -v amdgpu.ids $O/sweep_k3.txt
echo "== row_rate"; timeout -k 10 300 tools/row_rate > $O/row_rate.txt 2>&1; head -3 $O/row_rate.txt
