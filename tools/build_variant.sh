#!/bin/bash
# Tuning builds of libctcfa_hip.so from the current sources with extra -D macros (one vocabulary pitch:
# compiles in ~25 s).  Load one with CTCFA_ALLOW_TUNING_BUILD=1 CTCFA_LIB=$PWD/variants/<name>.so (the binding refuses tuning
# builds otherwise: ctcfa_build_flags()).
#   tools/build_variant.sh stamp3 -DCTCFA_STAMP=3        per-tile cycle stamps (tools/stamps2.py)
#   tools/build_variant.sh trace4 -DCTCFA_STAMP=4        per-group timeline of every tile (tools/trace4.py)
#   tools/build_variant.sh bb1 -DCTCFA_BODY_BLOCKS=1     one 32-row block per body of the tile loop (rounds 1-3)
#   tools/build_variant.sh btstamp -DCTCFA_BT_STAMP      cycle stamps of the checkpoint-mode backtrack (tools/bt_stamps.py)
#   tools/build_variant.sh pf3 -DCTCFA_PF=3              also CTCFA_POLL_LEAD, CTCFA_NBR_SLEEP, CTCFA_TWO_PROD32,
#                                                        CTCFA_VGPR_CAP, CTCFA_NO_DEADZONE, CTCFA_DEBUG_SPIN, CTCFA_TRACE_NT,
#                                                        CTCFA_SB_RING, CTCFA_SB_MARGIN
#   tools/build_variant.sh all                           stamp3 + trace4 + bb1 (what tools/fill_cycles.sh needs)
set -e
cd "$(dirname "$0")/.."
mkdir -p variants
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fno-slp-vectorize -DCTCFA_DEV_VP32_ONLY"
S=iterative-pseudo-forced-alignment-ctc_amd/csrc/ctcfa.hip
if [ "$1" = all ]; then
  /opt/rocm/bin/hipcc $F -DCTCFA_STAMP=3 $S -o variants/stamp3.so
  /opt/rocm/bin/hipcc $F -DCTCFA_STAMP=4 $S -o variants/trace4.so
  /opt/rocm/bin/hipcc $F -DCTCFA_BODY_BLOCKS=1 $S -o variants/bb1.so
else
  n=$1; shift
  /opt/rocm/bin/hipcc $F "$@" $S -o variants/$n.so
fi
