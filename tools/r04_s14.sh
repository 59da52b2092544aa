#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/s14
mkdir -p $O
echo "== windowed regime: round 3's library (a hipMalloc / hipFree of the table per call), then this round's (engine scratch)"
CTCFA_ALLOW_TUNING_BUILD=1 CTCFA_LIB=$PWD/variants/r3full.so timeout -k 10 400 python tools/windowed_timing.py 2>/dev/null | grep "T=" | tee $O/windowed_r3.txt
timeout -k 10 400 python tools/windowed_timing.py 2>/dev/null | grep "T=" | tee $O/windowed_r4.txt
