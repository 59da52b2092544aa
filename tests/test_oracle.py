"""CPU suite: the oracle (C restatement + NumPy twin) against each other, against
hand-derivable known answers, and against the committed golden vectors.

The golden vectors are restatement-generated (tests/golden/make_dp_goldens.py): the
reference holds no expected values for this path, so parity with the real
ctc-segmentation package is UNPINNED (see oracle/ctc_segmentation_oracle.c header).
"""
import os

import numpy as np
import pytest

from oracle import ctc_segmentation_twin as tw

GOLD = os.path.join(os.path.dirname(__file__), "golden", "dp_vectors.npz")
DUR = 320.4769 / 16000


def _twin(lpz, gt, ub, **kw):
    conf = tw.CtcSegmentationParameters(index_duration=DUR, **kw)
    tim, cp, sl = tw.ctc_segmentation(conf, lpz, np.asarray(gt).reshape(-1, 1))
    segs = tw.determine_utterance_segments(conf, list(ub), cp, tim, [""] * (len(ub) - 1))
    return tim, cp, sl, segs


@pytest.mark.parametrize("case", [(60, 8, 2, 5), (120, 32, 3, 10), (200, 32, 4, 20), (40, 5, 1, 30), (150, 29, 5, 9)])
def test_c_oracle_equals_numpy_twin(pkg, oracle, case):
    T, V, U, n = case
    lpz, gt, ub = pkg.synthetic.make_segment(sum(case), T, V, U, n)
    r = oracle.get_segments(lpz, gt, ub, oracle.make_config(index_duration=DUR))
    tim, cp, sl, segs = _twin(lpz, gt, ub)
    assert r["status"] == 0
    assert np.array_equal(tim, r["timings"])
    assert np.array_equal(cp, r["char_probs"])
    codes = [-2 if x == "" else (-1 if x == "ε" else int(x)) for x in sl]
    assert codes == r["state"].tolist()
    for (s, e, sc), a, b, c in zip(segs, r["seg_start"], r["seg_end"], r["seg_score"]):
        assert s == a and e == b and sc == c


@pytest.mark.parametrize("kw,okw", [
    (dict(preamble_transition_cost_zero=False), dict(preamble_transition_cost_zero=0)),
    (dict(backtrack_from_max_t=True), dict(backtrack_from_max_t=1)),
    (dict(blank_transition_cost_zero=True), dict(blank_transition_cost_zero=1)),
    (dict(score_min_mean_over_L=5), dict(score_min_mean_over_L=5)),
])
def test_config_variants_agree(pkg, oracle, kw, okw):
    lpz, gt, ub = pkg.synthetic.make_segment(77, 140, 32, 3, 12)
    r = oracle.get_segments(lpz, gt, ub, oracle.make_config(index_duration=DUR, **okw))
    tim, cp, _, segs = _twin(lpz, gt, ub, **kw)
    assert np.array_equal(tim, r["timings"]) and np.array_equal(cp, r["char_probs"])
    assert np.array_equal([s[2] for s in segs], r["seg_score"])


def test_windowed_regime_agrees(pkg, oracle):
    """T > min_window_size: per-column window offsets (small window to keep the twin fast)."""
    lpz, gt, ub = pkg.synthetic.make_segment(5, 260, 16, 3, 8)
    r = oracle.get_segments(lpz, gt, ub, oracle.make_config(index_duration=DUR, min_window_size=80, max_window_size=1000))
    conf = tw.CtcSegmentationParameters(index_duration=DUR, min_window_size=80, max_window_size=1000)
    try:
        tim, cp, _ = tw.ctc_segmentation(conf, lpz, gt.reshape(-1, 1))
        assert r["status"] == 0
        assert np.array_equal(tim, r["timings"]) and np.array_equal(cp, r["char_probs"])
    except IndexError:
        assert r["status"] == 2


def test_audio_shorter_than_text(pkg, oracle):
    lpz, gt, ub = pkg.synthetic.make_segment(1, 30, 32, 2, 20)
    assert len(gt) > 30
    assert oracle.get_segments(lpz, gt, ub)["status"] == oracle.AUDIO_SHORTER_THAN_TEXT
    with pytest.raises(AssertionError, match="Audio is shorter than text!"):
        tw.ctc_segmentation(tw.CtcSegmentationParameters(), lpz, gt.reshape(-1, 1))


def test_known_answer_forced_diagonal(pkg, oracle):
    """T == C: the only path switches on every frame -> frame_of_label = 0..C-1."""
    lpz, gt, ub = pkg.synthetic.make_segment(5, 62, 32, 2, 29)
    r = oracle.get_segments(lpz, gt, ub, oracle.make_config(index_duration=DUR))
    assert r["status"] == 0 and r["t_end"] == 61
    assert np.array_equal(r["frame_of_label"], np.arange(62))
    expect = np.concatenate([[0.0], lpz[np.arange(1, 62), gt[1:]].astype(np.float64)])
    assert np.array_equal(r["char_probs"], expect)


def test_known_answer_single_utterance(oracle):
    """[-1, blank, a, blank] on 4 frames with one-hot emissions: hand-derived path."""
    V = 4
    lpz = np.full((6, V), -20.0, np.float32)
    lpz[:, 0] = -0.01          # blank likely everywhere ...
    lpz[3, :] = -20.0
    lpz[3, 2] = -0.01          # ... but frame 3 says label 2
    gt = np.array([-1, 0, 2, 0])
    r = oracle.get_segments(lpz, gt, np.array([1, 3]), oracle.make_config(index_duration=0.02))
    assert r["status"] == 0
    assert r["frame_of_label"][2] == 3          # label 2 enters at frame 3
    assert r["frame_of_label"][3] == 4 and r["t_end"] == 4   # closing blank right after; best end cell there
    assert r["char_probs"][3] == np.float32(-0.01)


def test_path_properties(pkg, oracle):
    rng = np.random.default_rng(3)
    for seed in range(12):
        T = int(rng.integers(50, 400))
        U = int(rng.integers(1, 5))
        n = int(rng.integers(3, max(4, (T - 3) // (U + 1) // 2)))
        lpz, gt, ub = pkg.synthetic.make_segment(1000 + seed, T, 32, U, n)
        r = oracle.get_segments(lpz, gt, ub, oracle.make_config(index_duration=DUR))
        fol = r["frame_of_label"]
        assert r["status"] == 0
        assert fol[0] == 0 and (np.diff(fol[1:]) > 0).all(), "label frames must be strictly increasing"
        assert fol[-1] <= r["t_end"] < T
        visited = r["state"] != -2
        assert visited[1:r["t_end"] + 1].all() and not visited[0] and not visited[r["t_end"] + 1:].any()
        assert (r["char_probs"] <= 0).all() and (r["seg_score"] <= 0).all()
        assert (r["seg_end"] >= r["seg_start"]).all()


def test_np_mean_pairwise_matches_numpy(oracle):
    """Scores use NumPy's pairwise mean; the C restatement must be bit-identical to np.mean."""
    rng = np.random.default_rng(0)
    for L in (1, 5, 7, 8, 9, 30, 31, 64, 127, 128, 129, 300):
        T = 400
        cp = -rng.random(T) * 5
        # one utterance spanning [10, 10 + 2L + 3) so that both branches are exercised
        lpz = np.log(np.full((T, 3), 1 / 3, np.float32))
        gt = np.array([-1, 0, 1, 0])
        cfg = oracle.make_config(index_duration=1.0, score_min_mean_over_L=L)
        import ctypes
        tim = np.array([0.0, 10.0, 11.0, 10.0 + 2 * L + 3 + 1])
        ub = np.array([1, 3], np.int64)
        s, e, sc = np.zeros(1), np.zeros(1), np.zeros(1)
        P = lambda a, t: a.ctypes.data_as(ctypes.POINTER(t))
        rc = oracle.lib().oracle_determine_utterance_segments(
            ctypes.byref(cfg), P(ub, ctypes.c_int64), ctypes.c_int64(1), P(cp, ctypes.c_double), ctypes.c_int64(T),
            P(tim, ctypes.c_double), ctypes.c_int64(4), P(s, ctypes.c_double), P(e, ctypes.c_double), P(sc, ctypes.c_double))
        assert rc == 0
        conf = tw.CtcSegmentationParameters(index_duration=1.0, score_min_mean_over_L=L)
        ref = tw.determine_utterance_segments(conf, [1, 3], cp, tim, ["x"])
        assert (s[0], e[0], sc[0]) == ref[0]


def test_golden_vectors_pin_the_oracle(oracle):
    g = np.load(GOLD)
    assert "restatement" in str(g["provenance"])
    for name in g["names"]:
        blank, pre, maxt, L = (int(x) for x in g[f"{name}/cfg"])
        cfg = oracle.make_config(index_duration=float(g["index_duration"]), blank=blank,
                                 preamble_transition_cost_zero=pre, backtrack_from_max_t=maxt,
                                 score_min_mean_over_L=L)
        r = oracle.get_segments(g[f"{name}/lpz"], g[f"{name}/gt"], g[f"{name}/utt_begin"], cfg)
        assert r["status"] == int(g[f"{name}/status"]), name
        if r["status"] != 0:
            continue
        assert r["t_end"] == int(g[f"{name}/t_end"])
        assert np.array_equal(r["frame_of_label"], g[f"{name}/frame_of_label"]), name
        assert np.array_equal(r["char_probs"], g[f"{name}/char_probs"].astype(np.float64)), name
        assert np.array_equal(r["state"], g[f"{name}/state"]), name
        assert np.array_equal(r["seg_start"], g[f"{name}/seg_start"])
        assert np.array_equal(r["seg_end"], g[f"{name}/seg_end"])
        assert np.array_equal(r["seg_score"], g[f"{name}/seg_score"]), name


def test_prepare_token_list_prefix_property(pkg):
    """Labels of transcript[:k] are a prefix of the labels of the full transcript."""
    conf = tw.CtcSegmentationParameters()
    utts = [np.array([5, 6, 7]), np.array([8, 0, 9]), np.array([10])]
    full, ub = tw.prepare_token_list(conf, utts)
    assert full[:, 0].tolist() == [-1, 0, 5, 6, 7, 0, 8, 0, 9, 0, 10, 0] and ub == [1, 5, 9, 11]
    for k in (1, 2):
        part, ubk = tw.prepare_token_list(conf, utts[:k])
        assert np.array_equal(part[:, 0], full[: ub[k] + 1, 0]) and ubk == ub[: k + 1]
