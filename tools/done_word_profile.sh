#!/bin/bash
# Small calls with and without the completion word (CTCFA_NO_DONE_WORD=1: wait for the stream), and the replay workload.
set -o pipefail
mkdir -p gpurun_out/done
rm -rf gpurun_out/done/*
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/done/pytest.log 2>&1 || { tail -60 gpurun_out/done/pytest.log | cut -c1-300; exit 1; }
tail -2 gpurun_out/done/pytest.log
out=gpurun_out/done/r04_done_word.txt
for e in 0 1; do
  if [ $e = 1 ]; then export CTCFA_NO_DONE_WORD=1; echo "## CTCFA_NO_DONE_WORD=1 (the call waits for its stream)" >> $out; else unset CTCFA_NO_DONE_WORD; echo "## the host polls the completion word (shipped)" >> $out; fi
  timeout -k 10 300 python tools/call_latency.py 2>/dev/null | grep "T=" >> $out || exit 1
  timeout -k 10 200 python tools/call_trace.py 2>&1 | grep -v amdgpu.ids >> $out || exit 1
  timeout -k 10 300 python bench.py --workload replay --cpu-sample 0 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench.py --workload replay:', round(d['value'],1), d['unit'], d['ms_per_step'], 'ms per replay')" >> $out || exit 1
done
cat $out | cut -c1-220
