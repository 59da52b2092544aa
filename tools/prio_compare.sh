#!/bin/bash
# pipelined step time for two builds over batch shapes:  tools/prio_compare.sh libA.so libB.so
for cfg in "--utts 3 --utt-len 41" "--utts 6 --utt-len 41" "--utts 9 --utt-len 41" "--utts 15 --utt-len 33" "--utts 22 --utt-len 28" \
           "--utts 40 --utt-len 30" "--utts 59 --utt-len 25" "--segments 1024 --utts 6 --utt-len 41" "--segments 1024 --utts 22 --utt-len 28" \
           "--segments 4096 --utts 22 --utt-len 28" "--segments 4096 --utts 2 --utt-len 25 --frames 425" "--vocab 38" "--vocab 64"; do
  line="$cfg:"
  for L in "$@"; do
    CTCFA_LIB=$PWD/$L timeout -k 10 200 python bench.py --cpu-sample 0 --no-check --steps 300 --spinup-steps 500 $cfg > gpurun_out/mc.json 2>/dev/null || { echo "$cfg $L FAILED"; exit 1; }
    line="$line $(python -c "import json; d=json.load(open('gpurun_out/mc.json')); c=d['config']; r=d['roofline']; print('K%d/W%d %.4f (fill %.3f bt %.3f)' % (c['cols_per_lane'], c['waves_per_segment'], d['ms_per_step'], r['kernel_ms_avg'], r['backtrack_kernel_ms_avg']))") |"
  done
  echo "$line"
done
