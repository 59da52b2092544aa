#!/bin/bash
# Regenerate everything under profiles/ for one round on the MI355X box:
#   tools/collect_profiles.sh r03      (run from the repo root; writes gpurun_out/profiles_r03/; needs variants/btstamp.so:
#                                       tools/build_variant.sh btstamp -DCTCFA_BT_STAMP)
# Steps are joined so that a failing GPU step stops the rest.
R=${1:-r04}
PART=${2:-a}   # the whole collection does not fit one 20-minute gpurun call: part a, then part b
OUT=gpurun_out/profiles_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
if [ "$PART" = a ]; then
tools/collect_traffic.sh > $OUT/traffic.log 2>&1 && cp gpurun_out/pmc_traffic.json $OUT/${R}_pmc_traffic.json && cp gpurun_out/pmc_traffic.json profiles/${R}_pmc_traffic.json \
&& timeout -k 10 300 python bench.py > $OUT/${R}_bench.json 2> $OUT/bench.err \
&& timeout -k 10 300 python bench.py --serial > $OUT/${R}_bench_serial.json 2>> $OUT/bench.err \
&& timeout -k 10 300 python bench.py --workload replay > $OUT/${R}_bench_replay.json 2>> $OUT/bench.err \
&& timeout -k 10 300 python bench.py --workload replay --speculate 1 > $OUT/${R}_bench_replay_speculate1.json 2>> $OUT/bench.err \
&& timeout -k 10 300 python bench.py --workload replay --speculate 2 > $OUT/${R}_bench_replay_speculate2.json 2>> $OUT/bench.err \
&& timeout -k 10 300 python bench.py --workload words > $OUT/${R}_bench_words.json 2>> $OUT/bench.err \
&& timeout -k 10 300 python bench.py --workload corpus > $OUT/${R}_bench_corpus.json 2>> $OUT/bench.err \
&& rm -rf gpurun_out/kstats && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kstats -- python3 bench.py --steps 4000 --cpu-sample 0 > $OUT/kstats_bench.json 2> $OUT/kstats.err \
&& cp $(ls gpurun_out/kstats/*/*_kernel_stats.csv | head -1) $OUT/${R}_kernel_stats.csv \
&& rm -rf gpurun_out/kstats_s && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kstats_s -- python3 bench.py --steps 4000 --cpu-sample 0 --serial > $OUT/kstats_bench_serial.json 2>> $OUT/kstats.err \
&& cp $(ls gpurun_out/kstats_s/*/*_kernel_stats.csv | head -1) $OUT/${R}_kernel_stats_serial.csv \
&& tools/pmc_fill.sh "2" > $OUT/${R}_pmc_sq.txt 2>&1 \
&& (timeout -k 10 300 ./tools/row_rate) > $OUT/${R}_row_rate.txt 2>&1 \
&& python tools/reduce_issue.py $OUT/${R}_pmc_sq.txt $OUT/${R}_row_rate.txt > $OUT/${R}_fill_issue.json \
&& cp $OUT/${R}_fill_issue.json profiles/${R}_fill_issue.json \
&& timeout -k 10 300 python bench.py > $OUT/${R}_bench_with_issue.json 2>> $OUT/bench.err \
&& (for p in 0 1 2; do CTCFA_SB_PRIO=$p timeout -k 10 200 python tools/env_sweep.py "strider_prio$p" --steps 300 2>&1 | grep -v amdgpu.ids; done; for n in 2 4; do CTCFA_SB_WAVES=$n timeout -k 10 200 python tools/env_sweep.py "striders$n" --steps 300 2>&1 | grep -v amdgpu.ids; done) > $OUT/${R}_strider_knobs.txt 2>&1
else
timeout -k 10 600 python tools/sweep_shapes.py 2> $OUT/shapes.err | grep "^|" > $OUT/${R}_shapes.md \
&& timeout -k 10 300 python tools/vocab_sweep.py 2>/dev/null | grep "V=" > $OUT/${R}_vocab.txt \
&& timeout -k 10 300 python tools/call_latency.py 2>/dev/null | grep "T=" > $OUT/${R}_call_latency.txt \
&& timeout -k 10 300 python tools/windowed_timing.py 2>/dev/null | grep "T=" > $OUT/${R}_windowed.txt \
&& timeout -k 10 200 python tools/call_trace.py 2>&1 | grep -v amdgpu.ids > $OUT/${R}_call_trace.txt \
&& (timeout -k 10 200 python tools/small_modes.py 2>/dev/null; timeout -k 10 200 python tools/small_modes.py short 2>/dev/null; timeout -k 10 200 python tools/small_modes.py strider 2>/dev/null) | grep "T=" > $OUT/${R}_small_window_modes.txt \
&& (for n in 3 5; do echo "== striders: $n (+ 1 scoring wave); serial schedule, config 3"; CTCFA_SB_WAVES=$n CTCFA_ALLOW_TUNING_BUILD=1 CTCFA_LIB=$PWD/variants/btstamp.so timeout -k 10 100 python tools/bt_stamps.py 2>&1 | grep -v amdgpu.ids || exit 1; done; echo "== backtrack_from_max_t (94 blocks)"; FROM_MAX_T=1 CTCFA_ALLOW_TUNING_BUILD=1 CTCFA_LIB=$PWD/variants/btstamp.so timeout -k 10 100 python tools/bt_stamps.py 2>&1 | grep -v amdgpu.ids) > $OUT/${R}_backtrack_cycles.txt \
&& rm -rf gpurun_out/kt && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -- python3 bench.py --steps 600 --cpu-sample 0 --spinup-steps 200 > /dev/null 2>> $OUT/kstats.err \
&& (f=$(ls gpurun_out/kt/*/*_kernel_trace.csv | head -1); python tools/kernel_gaps.py $f fill_kernel; python tools/kernel_gaps.py $f stride_backtrack) > $OUT/${R}_kernel_gaps.txt && rm -rf gpurun_out/kt \
&& CTCFA_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --gpus 1 > $OUT/${R}_bench_rccl_one_rank.json 2>> $OUT/bench.err \
&& CTCFA_BENCH_REHEARSAL=1 timeout -k 10 300 python bench.py --gpus 2 --steps 200 > $OUT/${R}_bench_rehearsal_2ranks_gloo.json 2>> $OUT/bench.err \
&& tools/mode_compare.sh > $OUT/${R}_modes.txt 2>&1
fi
echo "collect_profiles part $PART exit $?"
