#!/usr/bin/env python3
"""Pin the oracle against the real package -- wherever ``ctc-segmentation==1.7.1`` can be imported.

The DP this repository rebuilds lives in a PyPI package the reference only pins
(/root/reference/requirements.txt:13); it is absent from the image this repository is built in, so
``oracle/ctc_segmentation_twin.py`` (laid out like the package, function by function) and the C oracle
restate it from its published algorithm: PARITY UNPINNED.
This script is the ten-minute job that pins it: run it in any environment that has the package,

    pip install ctc-segmentation==1.7.1 && python tools/diff_against_package.py

and it diffs the twin against the package function by function on the committed vectors
(tests/golden/dp_vectors.npz) plus windowed, multi-token and gratis-blank cases, and prints which of the
recalled details U1-U8 hold.  Nothing here runs in the product or in the test suites; it does not try to
obtain the package.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

try:
    import ctc_segmentation as pkg
except Exception as exc:   # the normal case in the build image
    print("ctc_segmentation is not importable here (%s: %s)." % (type(exc).__name__, exc))
    print("Nothing compared.  The oracle stays pinned only against itself, hand-derivable cases and the enumeration "
          "of tests/independent.py: parity unpinned.")
    sys.exit(2)

from oracle import ctc_segmentation_twin as twin   # noqa: E402

print("ctc_segmentation", getattr(pkg, "__version__", "?"), "from", os.path.dirname(pkg.__file__))
findings = {}


def note(key, ok, what):
    findings.setdefault(key, []).append((bool(ok), what))
    print(("  ok   " if ok else "  DIFF ") + key + ": " + what)


def both_configs(**kw):
    a, b = pkg.CtcSegmentationParameters(), twin.CtcSegmentationParameters()
    for k, v in kw.items():
        setattr(a, k, v)
        setattr(b, k, v)
    return a, b


def fill_both(lpz, gt, window, blank, flags):
    """cython_fill_table of both, on tables pre-filled the way ctc_segmentation() allocates them."""
    T, C = lpz.shape[0], gt.shape[0]
    W = min(window, T)
    out = []
    for mod in (pkg, twin):
        table = np.zeros([W, C], dtype=np.float32)
        table.fill(-10000000000.0)
        offsets = np.zeros([C], dtype=np.int64)
        fn = getattr(mod, "cython_fill_table", None)
        if fn is None:   # the package keeps it in its compiled submodule
            from ctc_segmentation.ctc_segmentation_dyn import cython_fill_table as fn
        t, c = fn(table, lpz.astype(np.float32), gt.astype(np.int64), offsets, blank, flags)
        out.append((table, offsets, int(t), int(c)))
    return out


def compare_case(name, lpz, gt, utt_begin, text_len, **cfg_kw):
    a, b = both_configs(**cfg_kw)
    gt2 = gt.reshape(-1, 1) if gt.ndim == 1 else gt
    # -- fill ------------------------------------------------------------------------------------
    window = a.min_window_size
    (ta, oa, tea, _), (tb, ob, teb, _) = fill_both(lpz, gt2, window, a.blank, a.flags)
    reach = ta > -1e9
    note("U1", np.array_equal(ta[reach], tb[reach]) and tea == teb, f"{name}: reachable table cells and end cell")
    note("U1", np.array_equal(ta, tb), f"{name}: every table cell, sentinels of unreachable cells included")
    if lpz.shape[0] > window:
        note("U2", np.array_equal(oa, ob), f"{name}: per-column window offsets")
    if a.flags & 1:
        note("U3", np.array_equal(ta[reach], tb[reach]), f"{name}: blank_transition_cost_zero table")
    if gt2.shape[1] > 1:
        note("U8", np.array_equal(ta[reach], tb[reach]), f"{name}: multi-character tokens, max over s")
    # -- backtrack ---------------------------------------------------------------------------------
    try:
        ra = pkg.ctc_segmentation(a, lpz.astype(np.float32), gt2.astype(np.int64))
    except Exception as exc:
        ra = exc
    try:
        rb = twin.ctc_segmentation(b, lpz.astype(np.float32), gt2.astype(np.int64))
    except Exception as exc:
        rb = exc
    if isinstance(ra, Exception) or isinstance(rb, Exception):
        note("backtrack", type(ra) is type(rb), f"{name}: both raise {type(ra).__name__} / {type(rb).__name__}")
        return
    note("backtrack", np.array_equal(ra[0], rb[0]), f"{name}: timings (residual rule, ties to STAY)")
    note("backtrack", np.array_equal(ra[1], rb[1]), f"{name}: char_probs")
    note("backtrack", list(ra[2]) == list(rb[2]), f"{name}: state_list")
    # -- utterance segments ----------------------------------------------------------------------
    if utt_begin is not None and len(utt_begin) > 1:
        text = ["x"] * (len(utt_begin) - 1)
        sa = pkg.determine_utterance_segments(a, utt_begin, ra[1], ra[0], text)
        sb = twin.determine_utterance_segments(b, utt_begin, rb[1], rb[0], text)
        note("U4", [s[:2] for s in sa] == [s[:2] for s in sb], f"{name}: utterance boundaries (int(round()), +-0.5 s clamp)")
        note("U4/U6", np.allclose([s[2] for s in sa], [s[2] for s in sb], rtol=0, atol=1e-12), f"{name}: scores (range bound, np.mean order)")


# ---- the committed vectors -------------------------------------------------------------------------
d = np.load(os.path.join(ROOT, "tests", "golden", "dp_vectors.npz"), allow_pickle=True)
dur = float(d["index_duration"])
for name in d["names"]:
    blank, preamble, maxt, L = (int(x) for x in d[name + "/cfg"])
    compare_case(str(name), d[name + "/lpz"], d[name + "/gt"], d[name + "/utt_begin"], None, blank=blank,
                 preamble_transition_cost_zero=bool(preamble), backtrack_from_max_t=bool(maxt), score_min_mean_over_L=L,
                 index_duration=dur)

# ---- regimes the production scripts do not reach, but the engine builds -----------------------------
import importlib   # noqa: E402
syn = importlib.import_module("iterative-pseudo-forced-alignment-ctc_amd.synthetic")
lpz, gt, ub = syn.make_segment(77, 900, 32, 6, 24)
compare_case("windowed (min_window_size 256)", lpz, gt, ub, None, min_window_size=256, index_duration=dur)
compare_case("gratis blank", lpz[:400], gt, ub, None, blank_transition_cost_zero=True, index_duration=dur)
rng = np.random.default_rng(5)
C, S = 60, 3
mat = np.full((C, S), -1, np.int64)
mat[1:, 0] = rng.integers(0, 32, C - 1)
for s in range(1, S):
    rows = rng.random(C) < 0.3
    rows[: s + 1] = False
    mat[rows, s] = rng.integers(0, 32, int(rows.sum()))
compare_case("multi-character tokens (S = 3)", lpz[:300], mat, None, None, index_duration=dur)

# ---- prepare_token_list / prepare_text (U5 is SpeechBrain's tokenizer path: not in this package) -----
a, b = both_configs()
tokens = [np.array([3, 4, 5]), np.array([], np.int64), np.array([0, 7]), np.array([9])]
ga, ua = pkg.prepare_token_list(a, [t.copy() for t in tokens])
gb, ub2 = twin.prepare_token_list(b, [t.copy() for t in tokens])
note("A.5", np.array_equal(ga, gb) and list(ua) == list(ub2), "prepare_token_list: label matrix and utterance starts")

print()
for key in sorted(findings):
    ok = all(o for o, _ in findings[key])
    print(f"{key:10s} {'holds' if ok else 'DIFFERS -- see the DIFF lines above'} ({sum(o for o, _ in findings[key])}/{len(findings[key])} checks)")
print("U5 (SpeechBrain tokenisation) and U7 (prefix property, a deduction the GPU tests check against independent runs) "
      "are outside this package.")
sys.exit(0 if all(o for v in findings.values() for o, _ in v) else 1)
