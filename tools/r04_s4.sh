#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/s4
mkdir -p $O
echo "== pytest" && timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
echo "== sweep (product build)"
timeout -k 10 400 python tools/env_sweep.py product "t0hi:CTCFA_TILE_PRIOS=3,2,2,3,3,3" "eq3:CTCFA_TILE_PRIOS=3" "p0:CTCFA_PROD_PRIO=0" "ns3:CTCFA_NS=3" product > $O/sweep_product.txt 2>&1; grep -v amdgpu.ids $O/sweep_product.txt
timeout -k 10 300 python tools/env_sweep.py b128 --segments 128 > $O/sweep_128.txt 2>&1; grep -v amdgpu.ids $O/sweep_128.txt
timeout -k 10 300 python tools/env_sweep.py b256 --segments 256 > $O/sweep_256.txt 2>&1; grep -v amdgpu.ids $O/sweep_256.txt
echo "== trace4"
CTCFA_ALLOW_TUNING_BUILD=1 CTCFA_LIB=$PWD/variants/trace4.so timeout -k 10 200 python tools/trace4.py 2 > $O/trace4_512.txt 2>&1; echo rc $?; grep -v amdgpu.ids $O/trace4_512.txt | head -14
CTCFA_ALLOW_TUNING_BUILD=1 CTCFA_LIB=$PWD/variants/trace4.so timeout -k 10 200 python tools/trace4.py 2 128 3000 22 28 > $O/trace4_128.txt 2>&1; echo rc $?; grep -v amdgpu.ids $O/trace4_128.txt | head -14
