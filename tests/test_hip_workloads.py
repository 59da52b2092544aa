"""GPU suite: the BASELINE.json configurations beyond configs[2] at their FULL sizes -- properties
that do not need the oracle over every segment, plus an oracle-checked sample (the oracle takes
seconds for the sample, minutes for the whole)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
DUR = 320.4769 / 16000
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _run(pkg, segs, **cfg):
    config = pkg.CtcSegmentationParameters(index_duration=DUR, **cfg)
    return pkg.ctc_segmentation.get_segments_device(config, [s[0] for s in segs], [s[1] for s in segs],
                                                    [s[2] for s in segs])


def _properties(segs, res):
    for (lpz, gt, ub), r in zip(segs, res):
        T, C = lpz.shape[0], len(gt)
        assert r["status"] == 0
        fol = r["frame_of_label"]
        assert 1 <= r["t_end"] <= T - 1 and fol[0] == 0
        assert np.all(np.diff(fol[1:]) > 0) and fol[1] >= 1 and fol[-1] <= r["t_end"]   # one switch per label column, in order
        cp = r["char_prob"]
        assert np.all(cp <= 0) and np.all(cp[r["t_end"] + 1:] == 0) and cp[0] == 0
        # a SWITCH frame reports the label's own emission
        assert np.array_equal(cp[fol[1:]], lpz[fol[1:], gt[1:]])
        assert np.all(r["seg_start"] <= r["seg_end"] + 1e-12)
        ok = r["seg_score"] > -1e9
        assert np.all(r["seg_score"][ok] <= 0)


def _same(a, b):
    return (a["status"] == b["status"] and a["t_end"] == b["t_end"] and np.array_equal(a["frame_of_label"], b["frame_of_label"])
            and np.array_equal(a["char_prob"], b["char_prob"]) and np.array_equal(a["seg_score"], b["seg_score"]))


def _check_sample(oracle, segs, res, idx):
    ocfg = oracle.make_config(index_duration=DUR)
    for i in idx:
        lpz, gt, ub = segs[i]
        o = oracle.get_segments(lpz, gt, ub, ocfg)
        r = res[i]
        assert r["status"] == o["status"] == 0 and r["t_end"] == o["t_end"]
        assert np.array_equal(r["frame_of_label"], o["frame_of_label"])
        assert np.array_equal(r["char_prob"].astype(np.float64), o["char_probs"])
        assert np.array_equal(r["seg_start"], o["seg_start"]) and np.array_equal(r["seg_end"], o["seg_end"])
        np.testing.assert_allclose(r["seg_score"], o["seg_score"], rtol=0, atol=1e-4)


def test_config3_ten_thousand_word_rows(pkg, oracle):
    """align_words over 10 000 utterances (configs[3]): every row's path is a valid monotone path;
    the result of a row does not depend on which rows share its launch (the path shards by rows);
    200 rows against the oracle."""
    segs = pkg.synthetic.make_word_rows(10000)
    res = _run(pkg, segs)
    _properties(segs, res)
    # the same rows in two rank-sized shards, in another order: identical per-row results
    order = np.random.default_rng(1).permutation(len(segs))
    for part in (order[:5000], order[5000:]):
        sub = _run(pkg, [segs[i] for i in part])
        assert all(_same(res[i], r) for i, r in zip(part, sub))
    _check_sample(oracle, segs, res, np.random.default_rng(2).choice(len(segs), 200, replace=False))


def test_config4_corpus_window_stream_one_gpu_share(pkg, oracle):
    """100 h corpus (configs[4]), the share of one of eight GPUs: 12.5 h of windows drawn from the
    recorded window sequence of the sample file (DP-only, synthetic emissions)."""
    calls = json.load(open(os.path.join(GOLD, "replay_windows.json")))["calls"]
    calls = [c for c in calls if c["C"] <= c["T"]]
    frames = int(100 * 3600 / DUR / 8)
    drawn = pkg.synthetic.draw_corpus_calls(calls, frames)
    assert sum(c["T"] for c in drawn) >= frames
    segs = pkg.synthetic.make_windows_like(drawn)
    res = []
    for k in range(0, len(segs), 2048):          # launches of 2048 windows
        res.extend(_run(pkg, segs[k:k + 2048]))
    _properties(segs, res)
    _check_sample(oracle, segs, res, np.random.default_rng(3).choice(len(segs), 200, replace=False))


def test_config1_replay_windows_dp_only(pkg, oracle):
    """The 183 recorded DP calls of the sample file as one ragged launch and one call at a time (what
    a single file's anchor iteration issues): identical results, all against the oracle."""
    calls = json.load(open(os.path.join(GOLD, "replay_windows.json")))["calls"]
    segs = pkg.synthetic.make_windows_like(calls, seed=1)
    res = _run(pkg, segs)
    ocfg = oracle.make_config(index_duration=DUR)
    for (lpz, gt, ub), r in zip(segs, res):
        o = oracle.get_segments(lpz, gt, ub, ocfg)
        assert r["status"] == o["status"]
        if o["status"] == 0:
            assert np.array_equal(r["frame_of_label"], o["frame_of_label"]) and r["t_end"] == o["t_end"]
            np.testing.assert_allclose(r["seg_score"], o["seg_score"], rtol=0, atol=1e-4)
    for i in range(0, len(segs), 7):
        one = _run(pkg, [segs[i]])[0]
        assert _same(one, res[i])
