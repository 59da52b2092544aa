#!/bin/bash
# usage: tools/sweep.sh "<lib list>" "<K list>"   -> one line per (lib, K)
for L in $1; do for K in $2; do
  CTCFA_ALLOW_TUNING_BUILD=1 CTCFA_LIB=$PWD/$L timeout -k 10 200 python bench.py --steps 300 --warmup 20 --cpu-sample 0 --cols-per-lane $K $SWEEP_FLAGS > gpurun_out/sw.json 2> gpurun_out/sw.err
  python - "$L" "$K" <<'PY'
import json,sys
try:
    d=json.load(open("gpurun_out/sw.json")); r=d["roofline"]
    print(sys.argv[1], "K",d["config"]["cols_per_lane"],"W",d["config"]["waves_per_segment"],"ms/step %.3f fill %.3f (min %.3f) bt %.3f frac %.4f"%(d["ms_per_step"],r["kernel_ms_avg"],r["kernel_ms_min"],r["backtrack_kernel_ms_avg"],r["frac"]))
except Exception as e:
    print(sys.argv[1], sys.argv[2], "ERR", e, open("gpurun_out/sw.err").read()[-400:])
PY
done; done
