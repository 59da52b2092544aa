#!/bin/bash
# GPU session 3: the rewritten tile loop (two-block bodies, lean hand-over): parity suite, timings of the variants, timeline
cd $GRAFT_REPO_ROOT
O=gpurun_out/s3
mkdir -p $O
echo "== pytest" && timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log
echo "== sweep (product build)"
timeout -k 10 300 python tools/env_sweep.py product "ns3:CTCFA_NS=3" > $O/sweep_product.txt 2>&1; grep -v amdgpu.ids $O/sweep_product.txt
for v in base bb1 poll2 poll6 pf3; do
  CTCFA_ALLOW_TUNING_BUILD=1 CTCFA_LIB=$PWD/variants/$v.so timeout -k 10 200 python tools/env_sweep.py $v > $O/sweep_$v.txt 2>&1; grep -v amdgpu.ids $O/sweep_$v.txt
done
echo "== trace4"
CTCFA_ALLOW_TUNING_BUILD=1 CTCFA_LIB=$PWD/variants/trace4.so timeout -k 10 200 python tools/trace4.py 2 --dump $O/trace4_512.npz > $O/trace4_512.txt 2>&1; echo rc $?; grep -v amdgpu.ids $O/trace4_512.txt | head -30
echo "== bench"
timeout -k 10 300 python bench.py --cpu-sample 0 > $O/bench.json 2> $O/bench.err; echo rc $?
timeout -k 10 300 python bench.py --cpu-sample 0 --serial > $O/bench_serial.json 2> $O/bench_serial.err; echo rc $?
timeout -k 10 300 python bench.py --cpu-sample 0 --vocab 38 > $O/bench_v38.json 2> $O/bench_v38.err; echo rc $?
timeout -k 10 300 python bench.py --cpu-sample 0 --vocab 64 > $O/bench_v64.json 2> $O/bench_v64.err; echo rc $?
python - <<'PY'
import json
for f in ("bench","bench_serial","bench_v38","bench_v64"):
    try:
        j=json.loads(open("gpurun_out/s3/%s.json"%f).read().strip().splitlines()[-1]); r=j["roofline"]
        print(f, "ms/step", round(j["ms_per_step"],4), "events", (j.get("ms_per_step_events") or {}).get("median"), "fill us", round(r["kernel_ms_avg"]*1e3,1), "bt us", round(r["backtrack_kernel_ms_avg"]*1e3,1), "value", round(j["value"]), j["config"].get("parity"))
    except Exception as e: print(f, "failed", e)
PY
