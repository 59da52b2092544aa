"""Diagnostic only: one small batch through the checkpoint-mode backtrack, compared with the oracle."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CTCFA_CHECKPOINT"] = "1"
import __graft_entry__ as ge
pkg = ge.build()
from oracle import oracle_c
syn = pkg.synthetic
V = int(os.environ.get("DBG_V", "32"))
cases = [(100, 1, 10), (200, 2, 20), (520, 1, 25), (900, 4, 28), (70, 2, 8)]
segs = [syn.make_segment(seed, T, V, U, n) for seed, (T, U, n) in enumerate(cases)]
cfg = pkg.CtcSegmentationParameters(index_duration=0.02)
res = pkg.ctc_segmentation.get_segments_device(cfg, [s[0] for s in segs], [s[1] for s in segs], [s[2] for s in segs])
ocfg = oracle_c.make_config(index_duration=0.02)
bad = 0
for k, ((lpz, gt, ub), r) in enumerate(zip(segs, res)):
    o = oracle_c.get_segments(lpz, gt, ub, ocfg)
    ok = (r["status"] == o["status"] and r["t_end"] == o["t_end"] and np.array_equal(r["frame_of_label"], o["frame_of_label"])
          and np.array_equal(r["char_prob"].astype(np.float64), o["char_probs"]))
    print("case", k, cases[k], "status", r["status"], o["status"], "t_end", r["t_end"], o["t_end"], "OK" if ok else "MISMATCH")
    if not ok:
        bad += 1
        a, b = np.asarray(r["frame_of_label"]), np.asarray(o["frame_of_label"])
        d = np.nonzero(a != b)[0]
        print("  fol hip   ", a[:40])
        print("  fol oracle", b[:40])
        print("  first diff at col", d[:5], "of", len(a))
sys.exit(1 if bad else 0)
