#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/s17
mkdir -p $O
for p in 0 1; do
 for w in "" "--workload words" "--workload corpus" "--workload replay"; do
  CTCFA_SB_PRIO=$p timeout -k 10 300 python bench.py --cpu-sample 0 $w > $O/b.json 2>/dev/null
  python - "$p" "$w" <<'PY'
import json,sys
j=json.loads(open("gpurun_out/s17/b.json").read().strip().splitlines()[-1]); r=j["roofline"]
print("strider prio", sys.argv[1], "|", sys.argv[2] or "synthetic", "| ms/step", round(j["ms_per_step"],4), "value", round(j["value"]), "fill us", round(r["kernel_ms_avg"]*1e3,1), "bt us", round(r["backtrack_kernel_ms_avg"]*1e3,1))
PY
 done
done 2>&1 | tee $O/workloads.txt
for v in 38 64 29; do for p in 0 1; do CTCFA_SB_PRIO=$p timeout -k 10 200 python tools/env_sweep.py "v${v}_prio$p" --vocab $v --steps 300 2>&1 | grep -v amdgpu.ids | cut -c1-215; done; done | tee $O/vocabs.txt
