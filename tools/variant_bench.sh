#!/bin/bash
# Tuning aid: serial fill time, pipelined step and a 128-segment batch for a list of library builds
# (variants/x_<name>.so).  usage: tools/variant_bench.sh name1 name2 ...
for n in "$@"; do
  for sh in "--serial" "" "--serial --segments 128"; do
    CTCFA_LIB=$PWD/variants/x_$n.so timeout -k 10 120 python bench.py --no-check --cpu-sample 0 --steps 400 --warmup 40 $sh 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$n', '[$sh]', 'ms/step', round(d['ms_per_step'],4), 'fill', round(r['kernel_ms_avg']*1e3,1), 'bt', round(r['backtrack_kernel_ms_avg']*1e3,1))" || exit 1
  done
done
