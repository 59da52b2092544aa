#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/s12
mkdir -p $O
echo "== pytest" && timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
echo "== call latency: results copied back (rounds 1-3) | written straight into the pinned result block"
CTCFA_NO_DIRECT_OUT=1 timeout -k 10 300 python tools/call_latency.py 2>/dev/null | grep "T=" > $O/call_latency_copy.txt; cat $O/call_latency_copy.txt
timeout -k 10 300 python tools/call_latency.py 2>/dev/null | grep "T=" > $O/call_latency_direct.txt; cat $O/call_latency_direct.txt
echo "== call trace"
CTCFA_NO_DIRECT_OUT=1 timeout -k 10 200 python tools/call_trace.py 2>&1 | grep -v amdgpu.ids > $O/call_trace_copy.txt; cat $O/call_trace_copy.txt
timeout -k 10 200 python tools/call_trace.py 2>&1 | grep -v amdgpu.ids > $O/call_trace_direct.txt; cat $O/call_trace_direct.txt
echo "== replay"
CTCFA_NO_DIRECT_OUT=1 timeout -k 10 300 python bench.py --workload replay --cpu-sample 0 > $O/replay_copy.json 2>/dev/null
timeout -k 10 300 python bench.py --workload replay --cpu-sample 0 > $O/replay_direct.json 2>/dev/null
python - <<'PY'
import json
for f in ("replay_copy","replay_direct"):
    j=json.loads(open("gpurun_out/s12/%s.json"%f).read().strip().splitlines()[-1]); print(f, "ms/step", round(j["ms_per_step"],3), "value", round(j["value"]))
PY
