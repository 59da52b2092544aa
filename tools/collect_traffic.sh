#!/bin/bash
# HBM traffic of the bench kernels from rocprofv3 PMC counters, as MI355X_MICROARCH.md §HBM
# prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (they do not fit one pass),
# unit KiB, and on gfx950 FETCH_SIZE counts exactly half the bytes of wide (16 B/lane)
# coalesced streaming reads -> doubled for the fill kernel, whose emission loads are dwordx4.
# Run on the MI355X box from the repo root:  tools/collect_traffic.sh  -> gpurun_out/pmc_traffic.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 6 --warmup 2 --spinup-steps 0 --cpu-sample 0 > gpurun_out/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 6 --warmup 2 --spinup-steps 0 --cpu-sample 0 > gpurun_out/pmc_write.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, collections, json
out = {}
for d, name in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    f = glob.glob(f"gpurun_out/{d}/*/*_counter_collection.csv")[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = "fill_kernel" if "fill_kernel" in r["Kernel_Name"] else ("backtrack_kernel" if "backtrack" in r["Kernel_Name"] else None)
        if k and r["Counter_Name"] == name:
            agg[k].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out.setdefault(k, {})[name + "_KiB_per_launch"] = sum(v) / len(v)
f = out["fill_kernel"]
f["hbm_bytes_per_launch"] = (2.0 * f["FETCH_SIZE_KiB_per_launch"] + f["WRITE_SIZE_KiB_per_launch"]) * 1024
f["correction"] = "FETCH_SIZE x2 (gfx950 counts 64 B per 128-B request on 16 B/lane streaming reads), WRITE_SIZE x1"
b = out["backtrack_kernel"]
b["hbm_bytes_per_launch_uncorrected"] = (b["FETCH_SIZE_KiB_per_launch"] + b["WRITE_SIZE_KiB_per_launch"]) * 1024
b["hbm_bytes_per_launch"] = b["hbm_bytes_per_launch_uncorrected"]
b["correction"] = "none: dword gathers, access width uncalibrated (a lower bound if its wide loads are under-counted like the fill's)"
out["schedule"] = "bench.py default schedule (backtrack(k) beside fill(k+1)); rocprofv3 counter collection serialises the dispatches"
out["workload"] = "bench.py defaults (512 x 3000 x 32, C=640), checkpoint mode"
json.dump(out, open("gpurun_out/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out))
PY
