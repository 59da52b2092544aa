#!/bin/bash
# Where the fill kernel's cycles go.  Needs the tuning builds variants/stamp3.so
# (-DCTCFA_STAMP=3) and variants/abl0..4.so (-DCTCFA_ABL=n: parts of the group hand-over left out, results
# WRONG, timing only), built from the current sources by tools/build_variant.sh all.
echo "## s_memtime stamps: per-tile totals and cycles per 8 rows of block 40 (config 3; then 128 segments)"
for a in "2" "2 128 3000 22 28"; do
  echo "== tools/stamps2.py $a"
  CTCFA_ALLOW_TUNING_BUILD=1 CTCFA_LIB=$PWD/variants/stamp3.so timeout -k 10 120 python tools/stamps2.py $a 2>&1 | grep -v amdgpu.ids || exit 1
done
echo "## hand-over ablation (bench.py --serial --no-check): 0 = product, 1 = no neighbour poll, 2 = + no halo read/select,"
echo "## 3 = + no exchange write, 4 = + counter once per block"
for L in 0 1 2 3 4; do
  for sh in "--segments 512" "--segments 128"; do
    CTCFA_ALLOW_TUNING_BUILD=1 CTCFA_LIB=$PWD/variants/abl$L.so timeout -k 10 120 python bench.py --serial --no-check --cpu-sample 0 --steps 300 --warmup 30 $sh 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('abl$L $sh: ms/step', round(d['ms_per_step'],4), 'fill us', round(r['kernel_ms_avg']*1e3,1), 'backtrack us', round(r['backtrack_kernel_ms_avg']*1e3,1))" || exit 1
  done
done
