#!/bin/bash
# GPU session 1 of round 4: tests, bench with rotating inputs vs one input set, timeline trace, priority sweep, windowed timing.
cd $GRAFT_REPO_ROOT
O=gpurun_out/s1
mkdir -p $O
echo "== pytest" && timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
echo "== bench (4 input sets)" && timeout -k 10 300 python bench.py > $O/bench_sets4.json 2> $O/bench_sets4.err; echo rc $?
echo "== bench (1 input set)" && timeout -k 10 300 python bench.py --input-sets 1 --cpu-sample 0 > $O/bench_sets1.json 2> $O/bench_sets1.err; echo rc $?
echo "== bench 20 steps" && timeout -k 10 300 python bench.py --steps 20 --cpu-sample 0 > $O/bench_20.json 2> $O/bench_20.err; echo rc $?
python - <<'PY'
import json
for f in ("sets4","sets1","20"):
    try:
        j=json.loads(open("gpurun_out/s1/bench_%s.json"%f).read().strip().splitlines()[-1]); r=j["roofline"]
        print(f, "ms/step", round(j["ms_per_step"],4), "events", j.get("ms_per_step_events"), "fill us", round(r["kernel_ms_avg"]*1e3,1), "bt us", round(r["backtrack_kernel_ms_avg"]*1e3,1), "value", round(j["value"]), j["config"].get("parity"))
    except Exception as e: print(f, "failed", e)
PY
echo "== trace4 (512 segments)"
CTCFA_ALLOW_TUNING_BUILD=1 CTCFA_LIB=$PWD/variants/trace4.so timeout -k 10 200 python tools/trace4.py 2 --dump $O/trace4_512.npz > $O/trace4_512.txt 2>&1; echo rc $?
CTCFA_ALLOW_TUNING_BUILD=1 CTCFA_LIB=$PWD/variants/trace4.so timeout -k 10 200 python tools/trace4.py 2 128 3000 22 28 --dump $O/trace4_128.npz > $O/trace4_128.txt 2>&1; echo rc $?
CTCFA_ALLOW_TUNING_BUILD=1 CTCFA_LIB=$PWD/variants/trace4tp.so timeout -k 10 200 python tools/trace4.py 2 --dump $O/trace4tp_512.npz > $O/trace4tp_512.txt 2>&1; echo rc $?
echo "== priority sweep (product build)"
timeout -k 10 600 python tools/env_sweep.py base "eq3:CTCFA_TILE_PRIOS=3" "eq2:CTCFA_TILE_PRIOS=2" "t0hi:CTCFA_TILE_PRIOS=3,2,2,3,3,3" "t01hi:CTCFA_TILE_PRIOS=3,3,2,2,3,3" \
   "lo4:CTCFA_TILE_PRIOS=2,2,2,2,3,3" "first3:CTCFA_TILE_PRIOS=3,3,3,2,2,2" "mid:CTCFA_TILE_PRIOS=2,3,3,3,3,2" "eq3p2:CTCFA_TILE_PRIOS=3;CTCFA_PROD_PRIO=2" "basep2:CTCFA_PROD_PRIO=2" "basep0:CTCFA_PROD_PRIO=0" \
   "ns3:CTCFA_NS=3" base > $O/prio_sweep.txt 2>&1; echo rc $?; cat $O/prio_sweep.txt | grep -v amdgpu.ids
echo "== priority sweep (two producers)"
CTCFA_ALLOW_TUNING_BUILD=1 CTCFA_LIB=$PWD/variants/twoprod.so timeout -k 10 400 python tools/env_sweep.py base "eq3:CTCFA_TILE_PRIOS=3" "t0hi:CTCFA_TILE_PRIOS=3,2,2,3,3,3" "eq3p2:CTCFA_TILE_PRIOS=3;CTCFA_PROD_PRIO=2" "basep2:CTCFA_PROD_PRIO=2" > $O/prio_sweep_twoprod.txt 2>&1; echo rc $?; cat $O/prio_sweep_twoprod.txt | grep -v amdgpu.ids
echo "== windowed"
timeout -k 10 400 python tools/windowed_timing.py > $O/windowed.txt 2>&1; echo rc $?; grep -v amdgpu.ids $O/windowed.txt
echo "== valu_rate2"; timeout -k 10 120 tools/valu_rate2 > $O/valu_rate2.txt 2>&1; cat $O/valu_rate2.txt
echo "== row_rate"; timeout -k 10 200 tools/row_rate > $O/row_rate.txt 2>&1; head -12 $O/row_rate.txt
echo "== pmc sq"; timeout -k 10 400 tools/pmc_fill.sh "2" > $O/pmc_sq.txt 2>&1; cat $O/pmc_sq.txt
