#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s33
rm -rf gpurun_out/s33/*
for a in "9500 40 30" "16000 100 29" "25000 200 29"; do
for e in 0 1; do
  if [ $e = 1 ]; then export CTCFA_BAND_ROW_BARRIER=1; else unset CTCFA_BAND_ROW_BARRIER; fi
  d=gpurun_out/s33/prof_$(echo $a | tr ' ' '_')_$e
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/windowed_one.py $a 4 > gpurun_out/s33/run.log 2>&1 || { tail -20 gpurun_out/s33/run.log; exit 1; }
  echo "== $a row_barrier=$e: $(grep 'call 3' gpurun_out/s33/run.log)"
  find $d -name "*kernel_stats.csv" | head -1 | xargs -I{} python3 -c "
import csv,sys
for r in csv.DictReader(open('{}')):
    if 'band' in r['Name'] or 'windowed' in r['Name']: print('   ', r['Name'][:48], r['Calls'], round(float(r['AverageNs'])/1e6,3), 'ms')
"
done; done
