#!/bin/bash
# Tuning aid: compile the library for one vocabulary pitch, print one kernel's resource usage and dump its ISA to /tmp/<name>.s
#   tools/devcompile.sh stride_backtrack [extra -D flags]
set -e
cd "$(dirname "$0")/../iterative-pseudo-forced-alignment-ctc_amd/csrc"
K=${1:-stride_backtrack}; shift || true
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -DCTCFA_DEV_VP32_ONLY $*"
/opt/rocm/bin/hipcc $F -fPIC -shared -Rpass-analysis=kernel-resource-usage ctcfa.hip -o /tmp/dev.so > /tmp/res.txt 2>&1 || { grep -m5 "error" /tmp/res.txt; exit 1; }
grep -A10 "Function Name: .*$K" /tmp/res.txt | grep -o "Function Name.*\|SGPRs: [0-9]*\|VGPRs: [0-9]*\|ScratchSize.*: [0-9]*\|Occupancy.*: [0-9]*\|Spill: [0-9]*" | tr '\n' ' '; echo
/opt/rocm/bin/hipcc $F --cuda-device-only -S ctcfa.hip -o /tmp/dev.s 2>/dev/null
awk -v k="$K" '$0 ~ "^_ZN5ctcfa[0-9]*"k"[^ ]*:" {f=1} f{print} /^\.Lfunc_end/{if(f){exit}}' /tmp/dev.s > /tmp/$K.s
echo "ISA: /tmp/$K.s ($(wc -l < /tmp/$K.s) lines, scratch ops: $(grep -c scratch_ /tmp/$K.s))"
