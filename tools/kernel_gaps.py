"""Diagnostic: gaps between consecutive launches of one kernel in a rocprofv3 --kernel-trace CSV.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -- python3 bench.py --steps 400 --cpu-sample 0
    python tools/kernel_gaps.py gpurun_out/kt/*/*_kernel_trace.csv fill_kernel"""
import csv, sys
import numpy as np
path, pat = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "fill_kernel"
rows = [r for r in csv.DictReader(open(path)) if pat in r["Kernel_Name"]]
s = np.array([int(r["Start_Timestamp"]) for r in rows]); e = np.array([int(r["End_Timestamp"]) for r in rows])
o = np.argsort(s); s, e = s[o], e[o]
n = len(s)
dur = (e - s)[n // 2:]; gap = (s[1:] - e[:-1])[n // 2:]; per = (s[1:] - s[:-1])[n // 2:]
print(f"{pat}: {n} launches; second half: duration median {np.median(dur)/1e3:.1f} us, gap to the next launch median {np.median(gap)/1e3:.1f} us "
      f"(p10 {np.percentile(gap,10)/1e3:.1f}, p90 {np.percentile(gap,90)/1e3:.1f}), start-to-start {np.median(per)/1e3:.1f} us")
