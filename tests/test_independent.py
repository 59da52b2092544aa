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
