#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s34
rm -rf gpurun_out/s34/*
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "window or spans or label_matri" > gpurun_out/s34/pytest.log 2>&1 || { tail -60 gpurun_out/s34/pytest.log | cut -c1-300; exit 1; }
tail -2 gpurun_out/s34/pytest.log
for a in "9500 40 30" "16000 100 29"; do
  d=gpurun_out/s34/prof_$(echo $a | tr ' ' '_')
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/windowed_one.py $a 4 > gpurun_out/s34/run.log 2>&1 || { tail -20 gpurun_out/s34/run.log; exit 1; }
  echo "== $a: $(grep 'call 3' gpurun_out/s34/run.log)"
  find $d -name "*kernel_stats.csv" | head -1 | xargs -I{} python3 -c "
import csv,sys
for r in csv.DictReader(open('{}')):
    if 'band' in r['Name'] or 'windowed' in r['Name']: print('   ', r['Name'][:48], r['Calls'], round(float(r['AverageNs'])/1e6,3), 'ms')
"
done
