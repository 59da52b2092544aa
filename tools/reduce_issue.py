#!/usr/bin/env python3
"""Reduce the fill kernel's SQ counters (tools/pmc_fill.sh output) and the isolated row loop's rates (tools/row_rate output)
to the JSON bench.py reads for `roofline.valu` / `roofline.issue_floor_ms`:
    python tools/reduce_issue.py profiles/r04_pmc_sq.txt profiles/r04_row_rate.txt > profiles/r04_fill_issue.json
SQ counters are summed over the chip by rocprofv3 (per launch here: the average over the run's fill launches); SQ_*_CYCLES and
SQ_ACTIVE_INST_* / SQ_WAIT_* count in units of 4 cycles per SIMD-wave on gfx9 -- the fractions below are ratios of counters of
the same kind, so the unit cancels."""
import json
import re
import sys

pmc, rr = sys.argv[1], sys.argv[2]
c = {}
wall = []
for line in open(pmc):
    if not line.startswith("K=2 "):
        continue
    m = re.search(r"fill wall (\d+)us", line)
    if m:
        wall.append(int(m.group(1)))
    for k, v in re.findall(r"(SQ_[A-Z_]+)=([0-9.e+]+)", line):
        c.setdefault(k, float(v))
valu = {
    "insts_per_launch": c.get("SQ_INSTS_VALU"),
    "salu_insts_per_launch": c.get("SQ_INSTS_SALU"),
    "lds_insts_per_launch": c.get("SQ_INSTS_LDS"),
    # a wave's life: issuing an instruction / parked at s_waitcnt, s_sleep or a barrier / ready but not issued
    "wave_issue_frac": c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"],
    "wave_parked_frac": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
    "wave_issue_stalled_frac": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
    "wave_valu_frac": c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"],
    # the vector ALUs of a SIMD: cycles some wave has a VALU instruction in flight / the SIMD's cycles in the launch.
    # SQ_BUSY_CYCLES is per shader engine (32 of them), SQ_ACTIVE_INST_VALU per wave -> normalise by the 1024 SIMDs.
    # (SQ_ACTIVE_INST_VALU counts about one unit per instruction here, whatever its length: the figure is built from the
    # instruction count and the measured issue cost of this kernel's mix instead -- 6 vector instructions of a K = 2 row take
    # 16.1 cycles of a SIMD alone, profiles/r01_issue_rates.txt / r04_valu_rate2.txt: 2.68 cycles an instruction)
    "busy_frac": (c["SQ_INSTS_VALU"] * 2.68 / 1024.0) / (c["SQ_BUSY_CYCLES"] / 32.0) if c.get("SQ_BUSY_CYCLES") else None,
    "busy_frac_note": "vector instructions per launch x 2.68 cycles (this row loop's measured mix) / (1024 SIMDs x the launch's cycles, SQ_BUSY_CYCLES per shader engine)",
    "lds_busy_frac": (c["SQ_LDS_IDX_ACTIVE"] / 256.0) / (c["SQ_BUSY_CYCLES"] / 32.0) if c.get("SQ_BUSY_CYCLES") and c.get("SQ_LDS_IDX_ACTIVE") else None,
    "lds_conflict_cycles_per_inst": c["SQ_LDS_BANK_CONFLICT"] / c["SQ_INSTS_LDS"] if c.get("SQ_INSTS_LDS") else None,
    "kernel_us_under_profiler": sum(wall) / len(wall) if wall else None,
    "counters": c,
}
cyc, ns = {}, {}
for line in open(rr):
    m = re.match(r"K=(\d+) mode=0:(.*)", line)
    if not m:
        continue
    for w, cy, n in re.findall(r"w(\d): +([0-9.]+) cyc/row +([0-9.]+) ns/row", m.group(2)):
        cyc["K%s_w%s" % (m.group(1), w)] = float(cy)
        ns["K%s_w%s" % (m.group(1), w)] = float(n)
json.dump({"source": [pmc, rr], "valu": valu, "row_rate": {"cycles_per_row": cyc, "ns_per_row": ns,
           "note": "tools/row_rate mode 0: the fill kernel's row loop (same ISA) alone, w = waves per SIMD"}}, sys.stdout, indent=1)
print()
