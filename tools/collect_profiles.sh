#!/bin/bash
# Regenerate everything under profiles/ for one round on the MI355X box:
#   tools/collect_profiles.sh r01      (run from the repo root; writes gpurun_out/profiles_r01/)
# Steps are joined so that a failing GPU step stops the rest.
R=${1:-r01}
OUT=gpurun_out/profiles_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
tools/collect_traffic.sh > $OUT/traffic.log 2>&1 && cp gpurun_out/pmc_traffic.json $OUT/${R}_pmc_traffic.json && cp gpurun_out/pmc_traffic.json profiles/${R}_pmc_traffic.json \
&& timeout -k 10 300 python bench.py > $OUT/${R}_bench.json 2> $OUT/bench.err \
&& timeout -k 10 300 python bench.py --serial > $OUT/${R}_bench_serial.json 2>> $OUT/bench.err \
&& rm -rf gpurun_out/kstats && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kstats -- python3 bench.py --steps 4000 --cpu-sample 0 > $OUT/kstats_bench.json 2> $OUT/kstats.err \
&& cp $(ls gpurun_out/kstats/*/*_kernel_stats.csv | head -1) $OUT/${R}_kernel_stats.csv \
&& rm -rf gpurun_out/kstats_s && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kstats_s -- python3 bench.py --steps 4000 --cpu-sample 0 --serial > $OUT/kstats_bench_serial.json 2>> $OUT/kstats.err \
&& cp $(ls gpurun_out/kstats_s/*/*_kernel_stats.csv | head -1) $OUT/${R}_kernel_stats_serial.csv \
&& timeout -k 10 600 python tools/sweep_shapes.py 2> $OUT/shapes.err | grep "^|" > $OUT/${R}_shapes.md \
&& timeout -k 10 300 python tools/windowed_timing.py 2>/dev/null | grep "T=" > $OUT/${R}_windowed.txt \
&& timeout -k 10 300 python tools/vocab_sweep.py 2>/dev/null | grep "V=" > $OUT/${R}_vocab.txt \
&& timeout -k 10 300 python tools/call_latency.py 2>/dev/null | grep "T=" > $OUT/${R}_call_latency.txt \
&& tools/pmc_lds.sh "2" > $OUT/${R}_pmc_lds.txt 2>&1 \
&& (timeout -k 10 120 ./tools/hwid_probe 6 58000; timeout -k 10 120 ./tools/hwid_probe 8 66000) > $OUT/${R}_wave_placement.txt 2>&1 \
&& (timeout -k 10 200 ./tools/valu_rate; timeout -k 10 120 ./tools/pk_rate; timeout -k 10 120 ./tools/lds_rate) > $OUT/${R}_issue_rates.txt 2>&1 \
&& timeout -k 10 400 python tools/wave_balance_test.py 2>/dev/null | grep "B=512" > $OUT/${R}_waves_per_simd.txt
echo "collect_profiles exit $?"
