#!/bin/bash
# Pipelined step time of bench.py in checkpoint mode and in decision-word mode over batch shapes:
# the measurements behind the mode rule of ctcfa_plan_create.
#   tools/mode_compare.sh > gpurun_out/rNN_modes.txt      (on the GPU box, from the repo root)
echo "# bench.py, pipelined schedule, vocab 32 unless stated. Per line: bench flags: then for checkpoint mode | decision-word mode | the plan's own choice:"
echo "#   segments x frames x label columns, tile shape, ms per step"
for cfg in "--utts 3 --utt-len 41" "--utts 6 --utt-len 41" "--utts 9 --utt-len 41" "--utts 15 --utt-len 33" \
           "--utts 18 --utt-len 30" "--utts 22 --utt-len 28" "--utts 40 --utt-len 30" "--utts 22 --utt-len 28 --frames 1000" \
           "--utts 22 --utt-len 28 --from-max-t" "--segments 128 --utts 22 --utt-len 28" "--segments 1024 --utts 6 --utt-len 41" \
           "--segments 1536 --utts 6 --utt-len 41" "--segments 2048 --utts 3 --utt-len 41" "--segments 4096 --utts 22 --utt-len 28" \
           "--segments 4096 --utts 2 --utt-len 25 --frames 425" "--vocab 38" "--vocab 64"; do
  line="$cfg:"
  for m in ck bits auto; do
    unset CTCFA_DECISION_BITS CTCFA_CHECKPOINT
    [ $m = ck ] && export CTCFA_CHECKPOINT=1
    [ $m = bits ] && export CTCFA_DECISION_BITS=1
    timeout -k 10 200 python bench.py --cpu-sample 0 --no-check --steps 300 --spinup-steps 500 $cfg > gpurun_out/mc.json 2>/dev/null || { echo "$cfg $m FAILED"; exit 1; }
    line="$line $(python -c "import json; d=json.load(open('gpurun_out/mc.json')); c=d['config']; print('%dx%dx%d K%d/W%d %.4f' % (c['segments_per_gpu'], c['frames'], c['label_columns'], c['cols_per_lane'], c['waves_per_segment'], d['ms_per_step']))") |"
  done
  echo "$line"
done
