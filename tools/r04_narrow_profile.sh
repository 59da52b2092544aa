#!/bin/bash
# Narrowed plans: config 3's shape with vocabularies of 29 .. 200 entries, texts over 28 of them --
# fill alone, backtrack alone, the pipelined step; plain plans against narrowed ones.  Then bench.py on the 38-entry case.
set -o pipefail
mkdir -p gpurun_out/narrow
out=gpurun_out/narrow/r04_narrowed.txt
rm -f $out
echo "## tools/env_sweep.py (512 x 3000 frames x 639 label columns, two input sets, texts over 28 entries)" >> $out
for v in 38 64 100 200; do
  CTCFA_NO_NARROW=1 timeout -k 10 200 python tools/env_sweep.py v${v}_plain --vocab $v --alphabet 28 2>&1 | grep -v amdgpu.ids >> $out || exit 1
  timeout -k 10 200 python tools/env_sweep.py v${v}_narrowed --vocab $v --alphabet 28 --narrow 2>&1 | grep -v amdgpu.ids >> $out || exit 1
done
for v in 32 29 20; do
  timeout -k 10 200 python tools/env_sweep.py v${v} --vocab $v 2>&1 | grep -v amdgpu.ids >> $out || exit 1
done
echo "## bench.py --vocab 38 --alphabet 28 (a narrowed plan: four input sets, parity gate on), then with CTCFA_NO_NARROW=1" >> $out
timeout -k 10 300 python bench.py --vocab 38 --alphabet 28 --cpu-sample 0 2>&1 | grep -v amdgpu.ids | tail -1 >> $out || exit 1
CTCFA_NO_NARROW=1 timeout -k 10 300 python bench.py --vocab 38 --alphabet 28 --cpu-sample 0 2>&1 | grep -v amdgpu.ids | tail -1 >> $out || exit 1
cat $out | cut -c1-260
