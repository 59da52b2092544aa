#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/s19
mkdir -p $O
(for v in base pace1 pace2 pf3 poll2 poll6 bb1 mask0 defer1 lean0 base; do
  CTCFA_ALLOW_TUNING_BUILD=1 CTCFA_LIB=$PWD/variants/$v.so timeout -k 10 200 python tools/env_sweep.py $v --steps 300 2>&1 | grep -v amdgpu.ids | cut -c1-215
done
CTCFA_ALLOW_TUNING_BUILD=1 CTCFA_LIB=$PWD/variants/base.so timeout -k 10 600 python tools/env_sweep.py "eq3:CTCFA_TILE_PRIOS=3" "t0hi:CTCFA_TILE_PRIOS=3,2,2,3,3,3" "lo4:CTCFA_TILE_PRIOS=2,2,2,2,3,3" "last:CTCFA_TILE_PRIOS=2,2,2,2,2,3" "p0:CTCFA_PROD_PRIO=0" "p2:CTCFA_PROD_PRIO=2" "ns3:CTCFA_NS=3" --steps 300 2>&1 | grep -v amdgpu.ids | cut -c1-215) | tee $O/knobs.txt
