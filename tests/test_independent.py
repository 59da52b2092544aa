"""Support for the unpinned oracle that does not depend on its decision rule:
the optimum of the recurrence by exhaustive path enumeration, against the C oracle (CPU suite) and
against the HIP path (GPU suite)."""
import numpy as np
import pytest

from tests import independent as ind


def _cases(seed, n, vocab=None):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        C = int(rng.integers(2, 7))
        T = int(rng.integers(C, 10))
        V = vocab or int(rng.integers(2, 6))
        out.append(ind.grid_case(rng, T, C, V))
    return out


def _assert_optimal(lpz, gt, res, preamble):
    best, t_end = ind.brute_force_optimum(lpz, gt, 0, preamble)
    assert res["status"] == 0
    assert res["t_end"] == t_end, (res["t_end"], t_end, best)
    cost = ind.path_from_result(lpz, gt, res["frame_of_label"], res["t_end"], 0, preamble)
    assert cost == best[t_end], (cost, best[t_end])
    # char_probs are the step costs along that path, except in the start column under
    # preamble_transition_cost_zero (the backtrack reports the blank posterior there, the table charges
    # nothing): from the first switch on, their exact sum is the optimum too
    cp = np.asarray(res["char_probs" if "char_probs" in res else "char_prob"], np.float64)
    first = int(res["frame_of_label"][1]) if preamble else 1
    assert float(np.sum(cp[first:t_end + 1])) == best[t_end]


@pytest.mark.parametrize("preamble", [True, False])
def test_oracle_path_attains_the_brute_force_optimum(oracle, preamble):
    cfg = oracle.make_config(index_duration=0.02, preamble_transition_cost_zero=int(preamble))
    for lpz, gt in _cases(101, 120):
        res = oracle.get_segments(lpz, gt, np.array([1, len(gt) - 1]) if len(gt) >= 3 else np.zeros(1, np.int64), cfg)
        _assert_optimal(lpz, gt, res, preamble)


def test_oracle_table_is_the_brute_force_table(oracle):
    """Every entry of the last column of the oracle's fill == the enumerated maximum."""
    for lpz, gt in _cases(202, 60):
        table, _, t_end = oracle.fill_table(lpz, gt, 10 ** 6, 0, 2)
        best, bt = ind.brute_force_optimum(lpz, gt)
        assert t_end == bt
        for t, b in enumerate(best):
            if b is not None:
                assert float(table[t, len(gt) - 1]) == b


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["auto", "checkpoint"])
@pytest.mark.parametrize("preamble", [True, False])
def test_hip_path_attains_the_brute_force_optimum(pkg, monkeypatch, mode, preamble):
    if mode == "checkpoint":
        monkeypatch.setenv("CTCFA_CHECKPOINT", "1")
    config = pkg.CtcSegmentationParameters(index_duration=0.02, preamble_transition_cost_zero=preamble)
    for V in (2, 3, 5):   # one launch per vocabulary size
        cases = _cases(303 + V, 120, vocab=V)
        res = pkg.ctc_segmentation.get_segments_device(config, [c[0] for c in cases], [c[1] for c in cases], None)
        for (lpz, gt), r in zip(cases, res):
            _assert_optimal(lpz, gt, r, preamble)


@pytest.mark.gpu
def test_hip_char_probs_sum_to_the_oracle_table_value(pkg, oracle):
    """On ordinary (non-grid) emissions: the fp32 running sum of char_prob along the returned path is
    the value the oracle's FILL (not its backtrack) reports for the end cell."""
    syn = pkg.synthetic
    segs = [syn.make_segment(4000 + s, T, 32, U, n) for s, (T, U, n) in
            enumerate([(300, 3, 20), (499, 4, 24), (150, 2, 12), (700, 6, 28), (1200, 10, 25), (64, 1, 10)])]
    config = pkg.CtcSegmentationParameters(index_duration=0.02)
    res = pkg.ctc_segmentation.get_segments_device(config, [s[0] for s in segs], [s[1] for s in segs], [s[2] for s in segs])
    for (lpz, gt, ub), r in zip(segs, res):
        table, _, t_end = oracle.fill_table(lpz, gt, 10 ** 6, 0, 2)
        assert r["t_end"] == t_end
        got = ind.sequential_fp32_sum(r["char_prob"], t_end, first=int(r["frame_of_label"][1]))
        assert got == pytest.approx(float(table[t_end, len(gt) - 1]), rel=2e-6, abs=1e-4)


def test_oracle_scoring_is_appendix_a4_with_numpys_own_mean(oracle, pkg):
    """determine_utterance_segments: the oracle's boundaries and scores (its restated pairwise summation
    included) against Appendix A.4 written with np.mean itself, on the oracle's own frames and char_probs."""
    syn = pkg.synthetic
    for L in (30, 7, 64):
        cfg = oracle.make_config(index_duration=0.0200298, score_min_mean_over_L=L)
        for s, (T, U, n) in enumerate([(300, 3, 20), (499, 4, 24), (150, 2, 12), (700, 6, 28), (1200, 10, 25), (64, 1, 10)]):
            lpz, gt, ub = syn.make_segment(5000 + s, T, 32, U, n)
            o = oracle.get_segments(lpz, gt, ub, cfg)
            st, en, sc = ind.utterance_segments(o["frame_of_label"], o["char_probs"], ub, 0.0200298, n=L)
            assert np.array_equal(o["seg_start"], st) and np.array_equal(o["seg_end"], en)
            np.testing.assert_allclose(o["seg_score"], sc, rtol=0, atol=1e-12)


@pytest.mark.gpu
def test_windowed_regime_never_beats_and_mostly_attains_the_brute_force_optimum(pkg):
    """The windowed regime (T > min_window_size: per-column windows of W rows that follow the running argmax)
    without the restatement, on grid cases with min_window_size 4..8 < T.  The package's windowed backtrack may
    leave its table (negative NumPy indices wrap around) and then describes no monotone path at all; whenever the
    result IS a monotone path through the trellis -- the frames it visited are the ones whose state is set -- no
    enumeration beats its exact cost, and in most cases (the window did not cut the optimum off) its cost is the
    enumerated maximum over the paths that end where it ends."""
    rng = np.random.default_rng(909)
    legal = attained = total = 0
    for m in (4, 5, 6, 7, 8):
        cases = []
        while len(cases) < 40:
            C = int(rng.integers(2, 6))
            T = int(rng.integers(max(C, m + 1), 10))
            cases.append(ind.grid_case(rng, T, C, 4))
        config = pkg.CtcSegmentationParameters(index_duration=0.02, min_window_size=m)
        res = pkg.ctc_segmentation.get_segments_device(config, [c[0] for c in cases], [c[1] for c in cases], None)
        for (lpz, gt), r in zip(cases, res):
            if r["status"] != 0:
                assert r["status"] == 2   # the package's IndexError after the last window doubling
                continue
            total += 1
            visited = np.nonzero(np.asarray(r["state"]) != -2)[0]
            if len(visited) == 0:
                continue
            t_last = int(visited.max())
            best, _ = ind.brute_force_optimum(lpz, gt)
            try:
                cost = ind.path_from_result(lpz, gt, r["frame_of_label"], t_last)
            except AssertionError:
                continue   # not a monotone path: index wrap-around (nothing to compare with)
            legal += 1
            assert best[t_last] is not None and cost <= best[t_last]
            attained += cost == best[t_last]
    assert total >= 150 and legal >= 0.6 * total and attained >= 0.7 * legal, (total, legal, attained)
