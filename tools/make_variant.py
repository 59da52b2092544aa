#!/usr/bin/env python3
"""Kernel-tuning helper (not product code): build ablated / instrumented copies of the HIP
library into variants/<name>.so from the CURRENT sources, by textual patches.

    python tools/make_variant.py base A C D stamp stamp+A ...

Patches (results are WRONG for everything except `base` and `stamp`; time them with
bench.py --no-check or tools/stamps.py):
    A      no cross-wave boundary LDS traffic in the row loop (no bnd read, no publish)
    C      no emission gathers (operands derived from registers)
    D      recurrence only: drop the residual/decision math (4 VALU per cell)
    nowait the pipelined entry's fill does not wait for the workspace hand-over (timing only)
    stamp  s_memtime stamps around compute / barrier of every step (see tools/stamps.py);
           host passes char_prob as the stamp buffer and skips the backtrack kernel
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "iterative-pseudo-forced-alignment-ctc_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fno-fast-math", "-fno-slp-vectorize"]


def sub(s, old, new, count=1):
    assert old in s, f"patch anchor not found: {old[:60]!r}"
    return s.replace(old, new, count)


def patch_A(k, h):
    k = sub(k, "                    *reinterpret_cast<float4*>(smem + out_addr + (i - 2) * 4) = pub4;", "", count=1)
    k = sub(k, "            float4 lin4 = *reinterpret_cast<const float4*>(smem + in_addr);",
            "            float4 lin4 = make_float4(kProbMax, kProbMax, kProbMax, kProbMax);", count=1)
    k = sub(k, "                    if (i + 4 < kRows) lin4_next = *reinterpret_cast<const float4*>(smem + in_addr + (i + 4) * 4);", "", count=1)
    return k, h


def patch_C(k, h):
    k = sub(k, "emq[d][k] = *reinterpret_cast<const float2*>(smem + gaddr[k] + d * (PITCH * 8));",
            "emq[d][k] = make_float2(-1.0f - 0.001f * (float)gaddr[k], -1.0f);")
    k = sub(k, "emq[i % PF][k] = *reinterpret_cast<const float2*>(smem + gaddr[k] + (i + PF) * (PITCH * 8));",
            "{}")
    return k, h


def patch_D(k, h):
    """decision-word mode only (run with CTCFA_DECISION_BITS=1)"""
    k = sub(k, """                    if constexpr (!CK) {
                        const float rsw = em[k].x - (nw - pl);
                        const float rst = em[k].y - (nw - pk);
                        // sign bit of (|rsw| - |rst|) == (|rst| > |rsw|): SWITCH; ties -> STAY
                        const float d = __builtin_fabsf(rsw) - __builtin_fabsf(rst);
                        dec[k] = __builtin_amdgcn_alignbit(dec[k], __float_as_uint(d), 31);
                    }""",
            "                    if constexpr (!CK) dec[k] ^= __float_as_uint(nw);")
    return k, h


def patch_nowait(k, h):
    """timing only (results race): the fill of run k does not wait for the backtrack of run k-2 to
    release its workspace -- shows how much of a pipelined step is that hand-over"""
    h = sub(h, "    if (pl->bt_pending[q] && hipEventQuery(pl->bt_done_ev[q]) != hipSuccess)\n        HIP_TRY(eng, hipStreamWaitEvent(st, pl->bt_done_ev[q], 0));", "")
    return k, h


def patch_G(k, h):
    """every gather reads entry 0 of its row (pure LDS broadcast: no bank conflicts, minimum data)"""
    k = sub(k, "gaddr[k] = static_cast<uint32_t>(lab) * 8u;", "gaddr[k] = 0u * static_cast<uint32_t>(lab);")
    return k, h


def patch_F(k, h):
    """regression aid: without the hand-over of a stopping tile's last row (the bug the fuzz test found)"""
    k = sub(k, "bnd[(w + 1) * kBndPitch + ((j + 1) * kRows) % kBnd] = pub4.x;", ";")
    return k, h


def patch_nobar(k, h):
    """timing only (results race): no workgroup barrier in the steady state of the fill kernel (compute
    tiles, V == pitch producer) -- upper bound of what neighbour-to-neighbour flags could gain"""
    k = sub(k, "        lds_barrier();\n    }\n    // row 32*nblk", "    }\n    // row 32*nblk")
    k = sub(k, "                lds_barrier();\n                continue;\n            }\n            jfirst = 0;", "                continue;\n            }\n            jfirst = 0;")
    k = sub(k, "                if (s + 1 < nblk) { vwrite(s + 1, ea); publish_flag(); }  // slot (s+1) % NS was last read in step s-1\n                lds_barrier();",
            "                if (s + 1 < nblk) { vwrite(s + 1, ea); publish_flag(); }\n                __builtin_amdgcn_s_sleep(40);")
    k = sub(k, "                if (s + 2 < nblk) { vwrite(s + 2, eb); publish_flag(); }\n                lds_barrier();\n            }\n        } else if constexpr (VP > 32 && VP <= 64) {",
            "                if (s + 2 < nblk) { vwrite(s + 2, eb); publish_flag(); }\n                __builtin_amdgcn_s_sleep(40);\n            }\n        } else if constexpr (VP > 32 && VP <= 64) {")
    return k, h


def patch_P(k, h):
    """backtrack kernel at wave priority 3 (above the fill kernel's 1/2)"""
    k = sub(k, "    if (sd.prestatus == kPreWindowed) return;  // windowed_kernel owns this segment", "    __builtin_amdgcn_s_setprio(3);\n    if (sd.prestatus == kPreWindowed) return;")
    return k, h


def patch_btwalk(k, h):
    """btstamp + split of the walk: cycles inside the switch loop go to seg_score[first utt]"""
    k, h = patch_btstamp(k, h)
    k = sub(k, "                    if (bb < kRows) {\n                        uint32_t tmp;", "                    const unsigned long long sw0 = __builtin_amdgcn_s_memtime();\n                    if (bb < kRows) {\n                        uint32_t tmp;")
    k = sub(k, "                    pc -= pidx;\n                    if (lane == 0) rec[j] = make_int2(cstart, (int)S);", "                    sw_acc += __builtin_amdgcn_s_memtime() - sw0;\n                    pc -= pidx;\n                    if (lane == 0) rec[j] = make_int2(cstart, (int)S);")
    k = sub(k, "            constexpr int kDepth = 4;", "            unsigned long long sw_acc = 0;\n            constexpr int kDepth = 4;")
    k = sub(k, "            if (pc - shift > 0) bad = 1;  // reached t == 0 in a label column: the package's IndexError", "            if (lane == 0 && seg_score) seg_score[sd.utt_off] = (double)sw_acc;\n            if (pc - shift > 0) bad = 1;")
    k = sub(k, "            seg_score[sd.utt_off + u] = min_avg;", "            if (u > 0) seg_score[sd.utt_off + u] = min_avg;")
    return k, h


PRIO_ANCHOR = "    if (KH != KL ? (my.role == kRoleHeavy) : (w >= (W + 1) / 2)) __builtin_amdgcn_s_setprio(2);\n    else __builtin_amdgcn_s_setprio(1);"


def patch_Q1(k, h):
    """heavy 3, light 1"""
    return sub(k, PRIO_ANCHOR, "    if (my.role == kRoleHeavy) __builtin_amdgcn_s_setprio(3);\n    else __builtin_amdgcn_s_setprio(1);"), h


def patch_Q2(k, h):
    """heavy 2, light 0 (with the producer)"""
    return sub(k, PRIO_ANCHOR, "    if (my.role == kRoleHeavy) __builtin_amdgcn_s_setprio(2);\n    else __builtin_amdgcn_s_setprio(0);"), h


def patch_Q4(k, h):
    """light tiles above heavy ones (3 / 2)"""
    return sub(k, PRIO_ANCHOR, "    if (my.role == kRoleHeavy) __builtin_amdgcn_s_setprio(2);\n    else __builtin_amdgcn_s_setprio(3);"), h


def patch_Q3(k, h):
    """heavy: later stages above earlier ones (3/2), light 1"""
    return sub(k, PRIO_ANCHOR, "    if (my.role != kRoleHeavy) __builtin_amdgcn_s_setprio(1);\n    else if (w >= 2) __builtin_amdgcn_s_setprio(3);\n    else __builtin_amdgcn_s_setprio(2);"), h


def patch_stamp(k, h):
    """s_memtime stamps of steps 40..55, slot (wave id, step, point): 0 = step start, 1 = before the
    barrier, 2 = after it.  Unreachable/dead-zone skipping is compiled out so that every wave works
    in every stamped step."""
    stamp_def = """
    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(seg_lastcol);
    auto stamp = [&](int s_, int q_) {
        unsigned long long t_;
        asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");
        const int slot_ = (wave_id * 16 + (s_ - 40)) * 3 + q_;
        if (lane == 0 && blockIdx.x < 8 && s_ >= 40 && s_ < 56) stamps[slot_] = t_;
    };
"""
    k = sub(k, "    if (my.role == kRoleProducer) {\n", stamp_def + "    if (my.role == kRoleProducer) {\n")
    k = sub(k, "        const int j = s - w;\n        if (j >= 0 && j < jfirst) {", "        stamp(s, 0);\n        const int j = s - w;\n        if (j >= 0 && j < jfirst) {")
    k = sub(k, "        lds_barrier();\n    }\n    // row 32*nblk", "        stamp(s, 1);\n        lds_barrier();\n        stamp(s, 2);\n    }\n    // row 32*nblk")
    k = sub(k, "                if (s + 2 < nblk) vload(s + 2, eb);\n                if (s + 1 < nblk) { vwrite(s + 1, ea); publish_flag(); }  // slot (s+1) % NS was last read in step s-1\n                lds_barrier();",
            "                stamp(s, 0);\n                if (s + 2 < nblk) vload(s + 2, eb);\n                if (s + 1 < nblk) { vwrite(s + 1, ea); publish_flag(); }\n                stamp(s, 1);\n                lds_barrier();\n                stamp(s, 2);")
    k = sub(k, "                if (s + 3 < nblk) vload(s + 3, ea);\n                if (s + 2 < nblk) { vwrite(s + 2, eb); publish_flag(); }\n                lds_barrier();",
            "                stamp(s + 1, 0);\n                if (s + 3 < nblk) vload(s + 3, ea);\n                if (s + 2 < nblk) { vwrite(s + 2, eb); publish_flag(); }\n                stamp(s + 1, 1);\n                lds_barrier();\n                stamp(s + 1, 2);")
    k = sub(k, "            if (w == wstar) {  // last-column scores", "            if (false) {  // last-column scores")
    k = sub(k, "                if (w == wstar) {\n                    const int t = j * kRows + lane;\n                    if (lane < kRows && t >= 1 && t < T) seg_lastcol[t] = kProbMax;\n                }", "")
    k = sub(k, "    if (w == wstar && lane == lstar && nblk * kRows < T) seg_lastcol[nblk * kRows] = pub4.x;", "")
    # stamps land in the caller's char_prob buffer; the backtrack kernel is not launched
    h = sub(h, "pl->d_segs, a.d_lpz, a.d_labels, pl->d_bits[ws], pl->d_lastcol[ws], pl->V, pl->prm.blank,",
            "pl->d_segs, a.d_lpz, a.d_labels, pl->d_bits[ws], a.d_char_prob, pl->V, pl->prm.blank,")
    h = sub(h, "    if ((rc = launch_backtrack(pl, a, want_seg, 0, st, ev ? ev[2] : nullptr, ev ? ev[3] : nullptr)) != CTCFA_OK) return rc;",
            "    (void)want_seg;")
    return k, h


def patch_btstamp(k, h):
    """backtrack kernel: cycles of phase 0 / A / B / C of segment b into status[b], t_end[b],
    seg_start[first utt], seg_end[first utt] (results are destroyed)."""
    k = sub(k, "    // ---- phase 0: first maximum of the last column", "    const unsigned long long bt0 = __builtin_amdgcn_s_memtime();\n    // ---- phase 0: first maximum of the last column")
    k = sub(k, "    // ---- phase A (wave 0): the walk", "    const unsigned long long bt1 = __builtin_amdgcn_s_memtime();\n    // ---- phase A (wave 0): the walk")
    k = sub(k, "    // ---- phase B: per-frame outputs, lanes = frames", "    const unsigned long long bt2 = __builtin_amdgcn_s_memtime();\n    // ---- phase B: per-frame outputs, lanes = frames")
    k = sub(k, "    if (!want_seg) return;\n    sync();", "    const unsigned long long bt3 = __builtin_amdgcn_s_memtime();\n    if (!want_seg) return;\n    sync();")
    k = sub(k, "    score_utterances<kThreads>(sd, p.L, p.dur, utt_begin + sd.utt_off + sd.seg_index, fol, cps, T, C, U,\n                               seg_start, seg_end, seg_score, tid, tick);\n}",
            "    score_utterances<kThreads>(sd, p.L, p.dur, utt_begin + sd.utt_off + sd.seg_index, fol, cps, T, C, U,\n                               seg_start, seg_end, seg_score, tid, tick);\n    sync();\n    const unsigned long long bt4 = __builtin_amdgcn_s_memtime();\n    if (tid == 0) { status_out[sd.seg_index] = (int)(bt1 - bt0); t_end_out[sd.seg_index] = (int)(bt2 - bt1); seg_start[sd.utt_off] = (double)(bt3 - bt2); seg_end[sd.utt_off] = (double)(bt4 - bt3); }\n}")
    return k, h


PATCHES = {"nobar": patch_nobar, "nowait": patch_nowait, "btwalk": patch_btwalk, "Q4": patch_Q4, "Q1": patch_Q1, "Q2": patch_Q2, "Q3": patch_Q3, "P": patch_P, "F": patch_F, "G": patch_G, "btstamp": patch_btstamp, "A": patch_A, "C": patch_C, "D": patch_D, "stamp": patch_stamp, "base": lambda k, h: (k, h)}


def main():
    os.makedirs(os.path.join(ROOT, "variants"), exist_ok=True)
    procs = []
    for name in sys.argv[1:]:
        defs = []
        parts = []
        for tok in name.split("+"):
            (defs if tok.startswith("-D") else parts).append(tok)
        k = open(os.path.join(CSRC, "ctcfa_kernels.hip.h")).read()
        h = open(os.path.join(CSRC, "ctcfa.hip")).read()
        for part in parts:
            k, h = PATCHES[part](k, h)
        d = os.path.join("/tmp", "ctcfa_variant_" + name.replace("+", "_").replace("=", "_"))
        shutil.rmtree(d, ignore_errors=True)
        os.makedirs(os.path.join(d, "p", "csrc"))
        os.makedirs(os.path.join(d, "include"))
        shutil.copy(os.path.join(ROOT, "include", "ctcfa.h"), os.path.join(d, "include"))
        open(os.path.join(d, "p", "csrc", "ctcfa_kernels.hip.h"), "w").write(k)
        open(os.path.join(d, "p", "csrc", "ctcfa.hip"), "w").write(h)
        out = os.path.join(ROOT, "variants", "exp_" + name.replace("+", "_").replace("=", "_") + ".so")
        procs.append((name, out, subprocess.Popen(["/opt/rocm/bin/hipcc"] + FLAGS + defs + ["ctcfa.hip", "-o", out],
                                                  cwd=os.path.join(d, "p", "csrc"), stderr=subprocess.PIPE)))
    for name, out, p in procs:
        err = p.communicate()[1].decode()
        print(name, "->", out if p.returncode == 0 else "FAILED\n" + err[-2000:])


if __name__ == "__main__":
    main()
