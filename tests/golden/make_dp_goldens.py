#!/usr/bin/env python3
"""Generate tests/golden/dp_vectors.npz: seeded inputs + expected outputs of the CTC
segmentation DP.

Provenance: ORACLE = RESTATEMENT.  The reference delegates this computation to the PyPI
package ctc-segmentation==1.7.1 (/root/reference/requirements.txt:13), which is not
installable here (no network) and whose results the reference's own tests never assert
(src/test/test_ctc_segmentation.py:40-43).  The expected outputs below therefore come from
oracle/ctc_segmentation_oracle.c (cross-checked against oracle/ctc_segmentation_twin.py),
NOT from the reference: they pin the oracle and the HIP path against regressions, they do
not pin either against the real package ("parity unpinned").

    python tests/golden/make_dp_goldens.py
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_c  # noqa: E402

syn = importlib.import_module("iterative-pseudo-forced-alignment-ctc_amd.synthetic")

CASES = [  # name, T, V, U, n, blank, config overrides
    ("cfg1_10s_window", 499, 32, 3, 34, 0, {}),           # BASELINE.json configs[0]: one 10 s VAD segment
    ("short_word", 97, 32, 1, 40, 0, {}),
    ("diag_T_eq_C", 62, 32, 2, 29, 0, {}),
    ("vocab29", 300, 29, 4, 18, 0, {}),
    ("blank31", 260, 32, 3, 20, 31, {"blank": 31}),
    ("no_preamble", 280, 32, 3, 18, 0, {"preamble_transition_cost_zero": 0}),
    ("from_max_t", 350, 32, 3, 22, 0, {"backtrack_from_max_t": 1}),
    ("score_L8", 400, 32, 4, 25, 0, {"score_min_mean_over_L": 8}),
    ("audio_shorter", 30, 32, 2, 20, 0, {}),              # C = 44 > T -> status 1
]
DUR = 320.4769 / 16000

out = {}
names = []
for i, (name, T, V, U, n, blank, kw) in enumerate(CASES):
    lpz, gt, ub = syn.make_segment(9000 + i, T, V, U, n, blank=blank)
    r = oracle_c.get_segments(lpz, gt, ub, oracle_c.make_config(index_duration=DUR, **kw))
    names.append(name)
    out[name + "/lpz"] = lpz
    out[name + "/gt"] = gt
    out[name + "/utt_begin"] = ub
    out[name + "/cfg"] = np.array([kw.get("blank", 0), kw.get("preamble_transition_cost_zero", 1),
                                   kw.get("backtrack_from_max_t", 0), kw.get("score_min_mean_over_L", 30)], np.int64)
    out[name + "/status"] = np.int64(r["status"])
    out[name + "/t_end"] = np.int64(r["t_end"])
    out[name + "/frame_of_label"] = r["frame_of_label"]
    out[name + "/char_probs"] = r["char_probs"].astype(np.float32)  # fp32-exact values
    out[name + "/state"] = r["state"]
    out[name + "/seg_start"] = r["seg_start"]
    out[name + "/seg_end"] = r["seg_end"]
    out[name + "/seg_score"] = r["seg_score"]
out["names"] = np.array(names)
out["index_duration"] = np.float64(DUR)
out["provenance"] = np.array("oracle: restatement (ctc-segmentation 1.7.1 unavailable)")
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dp_vectors.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes")
