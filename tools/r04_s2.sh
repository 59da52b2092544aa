#!/bin/bash
# GPU session 2 of round 4: sustained clocks (row loop alone vs the fill kernel), clean timeline, SCLK samples
cd $GRAFT_REPO_ROOT
O=gpurun_out/s2
mkdir -p $O
echo "== row_rate (sustained)"; timeout -k 10 300 tools/row_rate > $O/row_rate.txt 2>&1; head -6 $O/row_rate.txt
echo "== trace4 512"
CTCFA_ALLOW_TUNING_BUILD=1 CTCFA_LIB=$PWD/variants/trace4.so timeout -k 10 200 python tools/trace4.py 2 --dump $O/trace4_512.npz > $O/trace4_512.txt 2>&1; echo rc $?; grep -v amdgpu.ids $O/trace4_512.txt
CTCFA_ALLOW_TUNING_BUILD=1 CTCFA_LIB=$PWD/variants/trace4.so timeout -k 10 200 python tools/trace4.py 2 --timeline 0 > $O/trace4_512_tl.txt 2>&1; echo rc $?
echo "== trace4 128"
CTCFA_ALLOW_TUNING_BUILD=1 CTCFA_LIB=$PWD/variants/trace4.so timeout -k 10 200 python tools/trace4.py 2 128 3000 22 28 --dump $O/trace4_128.npz > $O/trace4_128.txt 2>&1; echo rc $?; grep -v amdgpu.ids $O/trace4_128.txt
echo "== sclk while the serial bench runs"
(timeout -k 10 120 python bench.py --serial --steps 60000 --cpu-sample 0 --input-sets 2 --no-check > $O/bench_serial_long.json 2> $O/bench_serial_long.err) &
BP=$!
sleep 25
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>&1 | grep -i -E "sclk|power|mclk" | head -6; sleep 1; done > $O/smi_serial.txt 2>&1
wait $BP; echo "bench rc $?"; cat $O/smi_serial.txt | head -20
python - <<'PY'
import json
j=json.loads(open("gpurun_out/s2/bench_serial_long.json").read().strip().splitlines()[-1]); r=j["roofline"]
print("serial long: ms/step", round(j["ms_per_step"],4), "fill us", round(r["kernel_ms_avg"]*1e3,1), "bt us", round(r["backtrack_kernel_ms_avg"]*1e3,1))
PY
echo "== sclk while the pipelined bench runs"
(timeout -k 10 120 python bench.py --steps 60000 --cpu-sample 0 --input-sets 2 --no-check > $O/bench_pipe_long.json 2> $O/bench_pipe_long.err) &
BP=$!
sleep 25
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>&1 | grep -i -E "sclk|power|mclk" | head -6; sleep 1; done > $O/smi_pipe.txt 2>&1
wait $BP; echo "bench rc $?"; cat $O/smi_pipe.txt | head -20
