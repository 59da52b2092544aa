#!/bin/bash
# Where the fill kernel's cycles go.  Needs the tuning builds of tools/build_variant.sh all:
# variants/stamp3.so (-DCTCFA_STAMP=3: per-tile totals, a stamp every 8 rows of one block), variants/trace4.so
# (-DCTCFA_STAMP=4: the per-group timeline of every tile) and variants/bb1.so (-DCTCFA_BODY_BLOCKS=1: one block per body).
echo "## s_memtime stamps: per-tile totals and cycles per 8 rows of block 40 (config 3; then 128 segments)"
for a in "2" "2 128 3000 22 28"; do
  echo "== tools/stamps2.py $a"
  CTCFA_ALLOW_TUNING_BUILD=1 CTCFA_LIB=$PWD/variants/stamp3.so timeout -k 10 120 python tools/stamps2.py $a 2>&1 | grep -v amdgpu.ids || exit 1
done
echo "## timeline (tools/trace4.py): per block rows 0-15 / rows 16-31 / to the next body's start, polls and waits, producer"
for a in "2" "2 128 3000 22 28"; do
  echo "== tools/trace4.py $a"
  CTCFA_ALLOW_TUNING_BUILD=1 CTCFA_LIB=$PWD/variants/trace4.so timeout -k 10 200 python tools/trace4.py $a 2>&1 | grep -v amdgpu.ids || exit 1
done
echo "## bodies of one block (rounds 1-3) against bodies of two (fill alone, pipelined step)"
for v in base bb1; do
  CTCFA_ALLOW_TUNING_BUILD=1 CTCFA_LIB=$PWD/variants/$v.so timeout -k 10 200 python tools/env_sweep.py $v 2>&1 | grep -v amdgpu.ids || exit 1
done
