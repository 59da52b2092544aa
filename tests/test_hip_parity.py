"""GPU parity: the HIP path (through the C ABI) against the oracle on seeded inputs.

Bar (BASELINE.json north_star): frame indices bit-exact, confidence scores within 1e-4.
char_prob values are fp32 emissions copied out, so they are compared exactly too.
"""
import numpy as np
import pytest

from tests import independent

pytestmark = pytest.mark.gpu

SCORE_TOL = 1e-4  # the tolerance north_star states for log-prob confidence scores
FUZZ_SALT = int(__import__("os").environ.get("CTCFA_FUZZ_SALT", "0"))  # soak runs: other random cases
DUR = 320.4769 / 16000


@pytest.fixture(autouse=True, params=["auto", "checkpoint"])
def dp_mode(request, monkeypatch):
    """Every test of this file runs twice: with the plan's own choice between decision-word mode
    and checkpoint mode (small test batches mostly get decision words), and with checkpoint mode
    forced wherever it exists (vocabularies up to 64 entries)."""
    if request.param == "checkpoint":
        monkeypatch.setenv("CTCFA_CHECKPOINT", "1")
    return request.param


def _check(pkg, oracle, segs, res, cfg_kw=None):
    ocfg = oracle.make_config(index_duration=DUR, **(cfg_kw or {}))
    for i, ((lpz, gt, ub), r) in enumerate(zip(segs, res)):
        o = oracle.get_segments(lpz, gt, ub, ocfg)
        assert r["status"] == o["status"], (i, r["status"], o["status"])
        if o["status"] != 0:
            continue
        assert r["t_end"] == o["t_end"], i
        assert np.array_equal(r["frame_of_label"], o["frame_of_label"]), f"segment {i}: frame indices differ"
        assert np.array_equal(r["char_prob"].astype(np.float64), o["char_probs"]), f"segment {i}: char_probs"
        assert np.array_equal(r["state"], o["state"]), f"segment {i}: state list"
        assert np.array_equal(r["seg_start"], o["seg_start"]), i
        assert np.array_equal(r["seg_end"], o["seg_end"]), i
        np.testing.assert_allclose(r["seg_score"], o["seg_score"], rtol=0, atol=SCORE_TOL)
        # ... and, without oracle/: Appendix A.4 with NumPy's own mean on the HIP path's own frames and char_probs
        if len(ub) > 1:
            st, en, sc = independent.utterance_segments(r["frame_of_label"], r["char_prob"], ub, DUR,
                                                        n=(cfg_kw or {}).get("score_min_mean_over_L", 30))
            assert np.array_equal(r["seg_start"], st) and np.array_equal(r["seg_end"], en), f"segment {i}: boundaries vs A.4"
            np.testing.assert_allclose(r["seg_score"], sc, rtol=0, atol=1e-12, err_msg=f"segment {i}: scores vs A.4 (np.mean)")


def _run(pkg, segs, **cfg):
    config = pkg.CtcSegmentationParameters(index_duration=DUR, **cfg)
    return pkg.ctc_segmentation.get_segments_device(config, [s[0] for s in segs], [s[1] for s in segs],
                                                    [s[2] for s in segs])


def test_ragged_batch_matches_oracle(pkg, oracle):
    syn = pkg.synthetic
    rng = np.random.default_rng(7)
    segs = []
    for seed in range(24):
        T = int(rng.integers(40, 900))
        U = int(rng.integers(1, 6))
        n = int(rng.integers(3, max(4, min(30, (T - 3) // (U + 1)))))
        segs.append(syn.make_segment(seed, T, 32, U, n))
    _check(pkg, oracle, segs, _run(pkg, segs))


def test_emissions_with_minus_infinity(pkg, oracle):
    """log(0) in the emissions (a masked vocabulary entry, an underflown posterior): sums stay at -inf or are lifted to
    max_prob by the recurrence's third operand, residuals become infinite -- the same IEEE arithmetic on both sides."""
    syn = pkg.synthetic
    segs = []
    for s, (T, U, n) in enumerate([(300, 3, 14), (700, 6, 20), (120, 1, 30), (1500, 10, 26)]):
        lpz, gt, ub = syn.make_segment(900 + s, T, 32, U, n)
        rng = np.random.default_rng(950 + s)
        lpz = lpz.copy()
        lpz[rng.random(lpz.shape) < 0.04] = -np.inf           # scattered entries
        lpz[:, int(rng.integers(1, 32))] = -np.inf             # one vocabulary entry masked everywhere
        lpz[int(rng.integers(1, T))] = -np.inf                 # one frame with nothing at all
        segs.append((lpz, gt, ub))
    _check(pkg, oracle, segs, _run(pkg, segs))


def test_emissions_that_are_not_log_posteriors_do_not_hang(pkg):
    """NaNs, +inf and positive numbers: whatever comes out (the package would not say anything sensible either), every
    segment answers with a status, and the engine serves the next call."""
    syn = pkg.synthetic
    bad = []
    for s, fill in enumerate([np.nan, np.inf, 50.0]):
        lpz, gt, ub = syn.make_segment(970 + s, 600, 32, 5, 20)
        lpz = lpz.copy()
        rng = np.random.default_rng(980 + s)
        lpz[rng.random(lpz.shape) < 0.1] = fill
        bad.append((lpz, gt, ub))
    res = _run(pkg, bad)
    assert len(res) == 3 and all(isinstance(int(r["status"]), int) for r in res)
    good = [syn.make_segment(990, 400, 32, 4, 20)]
    assert _run(pkg, good)[0]["status"] == 0


@pytest.mark.parametrize("K", [1, 2, 3, 4, 5, 6, 8, 10, 12, 16])
def test_every_lane_tile_width(pkg, oracle, engine, K):
    """Force each compiled cols-per-lane variant through the plan API (device buffers)."""
    import torch

    syn = pkg.synthetic
    segs = [syn.make_segment(100 + s, T, 32, U, n) for s, (T, U, n) in
            enumerate([(600, 5, 25), (333, 3, 17), (1000, 9, 30), (97, 1, 40)])]
    config = pkg.CtcSegmentationParameters(index_duration=DUR)
    T = [s[0].shape[0] for s in segs]
    C = [len(s[1]) for s in segs]
    U = [len(s[2]) - 1 for s in segs]
    plan = engine.plan(config.to_native(), 32, T, C, U, force_cols_per_lane=K)
    assert plan.info["cols_per_lane"] == K
    dev = torch.device("cuda:0")
    d_lpz = torch.from_numpy(np.concatenate([s[0].reshape(-1) for s in segs])).to(dev)
    d_lab = torch.from_numpy(np.concatenate([s[1] for s in segs]).astype(np.int32)).to(dev)
    d_ub = torch.from_numpy(np.concatenate([s[2] for s in segs]).astype(np.int32)).to(dev)
    nT, nC, nU = sum(T), sum(C), sum(U)
    d_fol = torch.empty(nC, dtype=torch.int32, device=dev)
    d_cp = torch.empty(nT, dtype=torch.float32, device=dev)
    d_st = torch.empty(nT, dtype=torch.int32, device=dev)
    d_seg = torch.empty(3, nU, dtype=torch.float64, device=dev)
    d_te = torch.empty(len(segs), dtype=torch.int32, device=dev)
    d_status = torch.empty(len(segs), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    plan.run_device(d_lpz.data_ptr(), d_lab.data_ptr(), d_ub.data_ptr(), d_fol.data_ptr(), d_cp.data_ptr(),
                    d_st.data_ptr(), d_seg[0].data_ptr(), d_seg[1].data_ptr(), d_seg[2].data_ptr(),
                    d_te.data_ptr(), d_status.data_ptr(), stream)
    torch.cuda.synchronize()
    fol, cp, st = d_fol.cpu().numpy(), d_cp.cpu().numpy(), d_st.cpu().numpy()
    seg, te, status = d_seg.cpu().numpy(), d_te.cpu().numpy(), d_status.cpu().numpy()
    res = []
    to = np.concatenate([[0], np.cumsum(T)])
    co = np.concatenate([[0], np.cumsum(C)])
    uo = np.concatenate([[0], np.cumsum(U)])
    for b in range(len(segs)):
        res.append(dict(status=int(status[b]), t_end=int(te[b]), frame_of_label=fol[co[b]:co[b + 1]],
                        char_prob=cp[to[b]:to[b + 1]], state=st[to[b]:to[b + 1]],
                        seg_start=seg[0][uo[b]:uo[b + 1]], seg_end=seg[1][uo[b]:uo[b + 1]],
                        seg_score=seg[2][uo[b]:uo[b + 1]]))
    _check(pkg, oracle, segs, res)
    plan.close()


@pytest.mark.parametrize("V", [5, 29, 32, 33, 38, 40, 45, 48, 52, 56, 57, 63, 64, 65, 76, 80, 81, 96, 97, 100, 112, 113, 128, 129,
                               150, 192, 193, 230, 256])
def test_vocabulary_sizes(pkg, oracle, V):
    syn = pkg.synthetic
    segs = [syn.make_segment(200 + s + V, T, V, U, n) for s, (T, U, n) in
            enumerate([(400, 4, 20), (250, 2, 31), (64, 1, 10)])]
    _check(pkg, oracle, segs, _run(pkg, segs))


@pytest.mark.parametrize("V", [33, 38, 48, 64, 100, 200, 256])
def test_narrowed_plans(pkg, oracle, V, monkeypatch):
    """A vocabulary above 32 entries whose texts use at most 31 labels each beside the blank (a character model's
    windows: the reference's 38-token model) runs through the 32-entry fill kernel on the columns each segment looks at
    (a NARROWED plan; the host-buffer entry narrows by itself): same results as the oracle, as the un-narrowed plan,
    with every flag, with the blank elsewhere, and a batch with one text of more than 31 labels is simply not narrowed."""
    syn = pkg.synthetic
    shapes = [(400, 4, 20), (250, 2, 31), (64, 1, 10), (700, 6, 25), (1100, 9, 30)]
    segs = [syn.make_segment(700 + s + V, T, V, U, n, alphabet=28 if s % 2 == 0 else 31) for s, (T, U, n) in enumerate(shapes)]
    assert all(len(np.unique(g[1:])) <= 32 for _, g, _ in segs)
    res = _run(pkg, segs)
    _check(pkg, oracle, segs, res)
    monkeypatch.setenv("CTCFA_NO_NARROW", "1")
    plain = _run(pkg, segs)
    monkeypatch.delenv("CTCFA_NO_NARROW")
    for a, b in zip(res, plain):
        for k in ("frame_of_label", "char_prob", "state", "seg_start", "seg_end", "seg_score", "t_end", "status"):
            assert np.array_equal(a[k], b[k]), k
    # flags: the start column pays for staying; the blank's stay step is free (checkpoint mode only: a narrowed plan's
    # backtrack stages the same 32 entries, whatever the vocabulary)
    _check(pkg, oracle, segs, _run(pkg, segs, preamble_transition_cost_zero=False), dict(preamble_transition_cost_zero=0))
    _check(pkg, oracle, segs, _run(pkg, segs, blank_transition_cost_zero=True), dict(blank_transition_cost_zero=1))
    _check(pkg, oracle, segs, _run(pkg, segs, backtrack_from_max_t=True), dict(backtrack_from_max_t=1))
    # the blank somewhere else in the vocabulary (texts over entries 0 .. 27 then)
    b = V - 1
    moved = [syn.make_segment(900 + s + V, T, V, U, n, blank=b, alphabet=28) for s, (T, U, n) in enumerate(shapes[:3])]
    _check(pkg, oracle, moved, _run(pkg, moved, blank=b), dict(blank=b))
    # a narrowed batch with a window of the windowed regime in it (its own kernels, the vocabulary as it is)
    if V in (38, 200):
        mixed = segs[:3] + [syn.make_segment(960 + V, 900, V, 10, 14, alphabet=28)]
        kw = dict(min_window_size=720, max_window_size=4000)
        _check(pkg, oracle, mixed, _run(pkg, mixed, **kw), cfg_kw=kw)
    # one text of more than 31 labels in the batch: no narrowing, same answers
    if V >= 38:
        wide = segs[:2] + [syn.make_segment(990 + V, 1500, V, 12, 40)]
        assert len(np.unique(wide[-1][1][1:])) > 32
        _check(pkg, oracle, wide, _run(pkg, wide))


@pytest.mark.parametrize("V", [38, 100, 256])
@pytest.mark.parametrize("how", ["labels", "promise"])
def test_narrowed_plan_through_the_plan_api(pkg, oracle, engine, V, how):
    """A narrowed plan for device-resident runs -- created with the labels (ctcfa_plan_create_shared looks at them) or with
    CTCFA_FLAG_TEXTS_OF_31_LABELS and no labels at all -- serial and pipelined, against the oracle; the same plan then serves
    OTHER texts of the same geometry (the segments' rings are derived on the device, run by run)."""
    import torch
    syn = pkg.synthetic
    B = 24
    config = pkg.CtcSegmentationParameters(index_duration=DUR)
    ocfg = oracle.make_config(index_duration=DUR)
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    plan = None
    for seed0 in (40, 140):
        segs = [syn.make_segment(seed0 + s, 600 + 20 * s, V, 5, 22, alphabet=27) for s in range(B)]
        T, C, U = [s[0].shape[0] for s in segs], [len(s[1]) for s in segs], [len(s[2]) - 1 for s in segs]
        labels = np.concatenate([s[1] for s in segs]).astype(np.int32)
        if plan is None:
            plan = (engine.plan(config.to_native(), V, T, C, U, labels=labels) if how == "labels"
                    else engine.plan(config.to_native(), V, T, C, U, texts_of_31_labels=True))
            assert plan.info["vocab_pitch"] == 34   # the 32-entry ring
        d_lpz = torch.from_numpy(np.concatenate([s[0].reshape(-1) for s in segs])).to(dev)
        d_lab = torch.from_numpy(labels).to(dev)
        d_ub = torch.from_numpy(np.concatenate([s[2] for s in segs]).astype(np.int32)).to(dev)
        ref = [oracle.get_segments(*s, ocfg) for s in segs]
        co = np.concatenate([[0], np.cumsum(C)])
        fo = np.concatenate([[0], np.cumsum(T)])
        for pipelined in (False, True, True):
            o = dict(fol=torch.zeros(sum(C), dtype=torch.int32, device=dev), cp=torch.zeros(sum(T), dtype=torch.float32, device=dev),
                     st=torch.zeros(sum(T), dtype=torch.int32, device=dev),
                     seg=torch.zeros(3, sum(U), dtype=torch.float64, device=dev), te=torch.zeros(B, dtype=torch.int32, device=dev),
                     status=torch.full((B,), -7, dtype=torch.int32, device=dev))
            plan.run_device(d_lpz.data_ptr(), d_lab.data_ptr(), d_ub.data_ptr(), o["fol"].data_ptr(), o["cp"].data_ptr(), o["st"].data_ptr(),
                            o["seg"][0].data_ptr(), o["seg"][1].data_ptr(), o["seg"][2].data_ptr(), o["te"].data_ptr(),
                            o["status"].data_ptr(), stream, pipelined=pipelined)
            plan.flush(stream)
            torch.cuda.synchronize()
            assert (o["status"].cpu().numpy() == 0).all()
            fol, cp, te, st = o["fol"].cpu().numpy(), o["cp"].cpu().numpy(), o["te"].cpu().numpy(), o["st"].cpu().numpy()
            for b, r in enumerate(ref):
                assert te[b] == r["t_end"]
                assert np.array_equal(fol[co[b]:co[b + 1]], r["frame_of_label"]), b
                assert np.array_equal(cp[fo[b]:fo[b + 1]].astype(np.float64), r["char_probs"]), b
                assert np.array_equal(st[fo[b]:fo[b + 1]], r["state"]), b
    plan.close()


@pytest.mark.parametrize("V", [65, 100, 200, 256])
def test_narrowed_plans_with_a_ring_of_64_entries(pkg, oracle, V, monkeypatch, engine):
    """Texts of 32 .. 62 labels beside the blank over a vocabulary of more than 64 entries: the 64-entry kernels on the
    columns each segment looks at (checkpoint mode where the plain plan has decision words only) -- same results as the
    oracle and as the plain plan, with the flags, with the blank elsewhere; 63 labels are one too many (plain plan)."""
    import torch
    syn = pkg.synthetic
    shapes = [(400, 4, 20), (700, 8, 31), (64, 1, 10), (1100, 12, 30), (1500, 14, 28)]
    segs = [syn.make_segment(1700 + s + V, T, V, U, n, alphabet=(50, 62, 40, 45, 58)[s]) for s, (T, U, n) in enumerate(shapes)]
    assert max(len(np.unique(g[1:])) for _, g, _ in segs) > 33
    res = _run(pkg, segs)
    _check(pkg, oracle, segs, res)
    monkeypatch.setenv("CTCFA_NO_NARROW", "1")
    plain = _run(pkg, segs)
    monkeypatch.delenv("CTCFA_NO_NARROW")
    for a, b in zip(res, plain):
        for k in ("frame_of_label", "char_prob", "state", "seg_start", "seg_end", "seg_score", "t_end", "status"):
            assert np.array_equal(a[k], b[k]), k
    _check(pkg, oracle, segs, _run(pkg, segs, preamble_transition_cost_zero=False), dict(preamble_transition_cost_zero=0))
    _check(pkg, oracle, segs, _run(pkg, segs, blank_transition_cost_zero=True), dict(blank_transition_cost_zero=1))
    _check(pkg, oracle, segs, _run(pkg, segs, backtrack_from_max_t=True), dict(backtrack_from_max_t=1))
    b = V - 1
    moved = [syn.make_segment(1900 + s + V, T, V, U, n, blank=b, alphabet=55) for s, (T, U, n) in enumerate(shapes[:3])]
    _check(pkg, oracle, moved, _run(pkg, moved, blank=b), dict(blank=b))
    # through the plan API (the labels given): the 64-entry ring, pitch 64 + 2
    T, C, U = [s[0].shape[0] for s in segs], [len(s[1]) for s in segs], [len(s[2]) - 1 for s in segs]
    config = pkg.CtcSegmentationParameters(index_duration=DUR)
    labels = np.concatenate([s[1] for s in segs]).astype(np.int32)
    plan = engine.plan(config.to_native(), V, T, C, U, labels=labels)
    assert plan.info["vocab_pitch"] == 66
    plan.close()
    if V >= 100:   # 63 labels in one text: no ring takes it
        wide = segs[:2] + [syn.make_segment(1990 + V, 2500, V, 20, 40, alphabet=63)]
        if len(np.unique(wide[-1][1][1:])) >= 64:
            plan = engine.plan(config.to_native(), V, [s[0].shape[0] for s in wide], [len(s[1]) for s in wide],
                               [len(s[2]) - 1 for s in wide], labels=np.concatenate([s[1] for s in wide]).astype(np.int32))
            assert plan.info["vocab_pitch"] != 66
            plan.close()
        _check(pkg, oracle, wide, _run(pkg, wide))


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_narrowed_plans(pkg, oracle, seed):
    """Random vocabularies (33 .. 256 entries), alphabets (2 .. 62 labels: the 32-entry ring, the 64-entry ring, or none),
    blanks, flags and ragged batches through the host-buffer entry against the oracle."""
    rng = np.random.default_rng(77_000 + seed)
    syn = pkg.synthetic
    for _ in range(4):
        V = int(rng.integers(33, 257))
        blank = int(rng.integers(0, V))
        alphabet = int(rng.integers(2, min(62, V - 1) + 1))
        kw = dict(blank=blank)
        if rng.random() < 0.3:
            kw["preamble_transition_cost_zero"] = False
        if rng.random() < 0.3:   # (at most 63 distinct labels: a ring, the vocabulary's own kernels, or the compact matrix take the flag)
            kw["blank_transition_cost_zero"] = True
        if rng.random() < 0.3:
            kw["backtrack_from_max_t"] = True
        segs = []
        for s in range(int(rng.integers(1, 7))):
            U = int(rng.integers(1, 9))
            n = int(rng.integers(2, 33))
            T = int(rng.integers(U * (n + 1) + 2, 3 * U * (n + 1) + 400))
            segs.append(syn.make_segment(int(rng.integers(1 << 30)), T, V, U, n, blank=blank, alphabet=alphabet))
        _check(pkg, oracle, segs, _run(pkg, segs, **kw), cfg_kw=kw)


def test_narrowed_plan_reports_a_text_of_more_than_31_labels(pkg, oracle, engine):
    """CTCFA_FLAG_TEXTS_OF_31_LABELS is the caller's promise: a segment whose text breaks it gets status
    CTCFA_ST_TOO_MANY_LABELS (zeroed outputs), the other segments of the run are aligned -- 31 labels beside the blank fit,
    32 do not; a blank among the labels does not count."""
    import torch
    syn = pkg.synthetic
    V, n = 64, 120
    rng = np.random.default_rng(5)

    def text(k):   # n labels over k distinct entries (all of them used), and some blanks
        g = np.concatenate([np.arange(1, k + 1), rng.integers(1, k + 1, size=n - k - 8), np.zeros(8, np.int64)])
        rng.shuffle(g)
        return np.concatenate([[-1], g]).astype(np.int64), np.array([1, 40, 80, n], dtype=np.int64)

    segs = []
    for s, k in enumerate((31, 32, 12, 40, 31)):
        gt, ub = text(k)
        segs.append((syn.make_emissions(np.random.default_rng(900 + s), 700, V, gt, 0), gt, ub))
    B = len(segs)
    T, C, U = [s[0].shape[0] for s in segs], [len(s[1]) for s in segs], [len(s[2]) - 1 for s in segs]
    config = pkg.CtcSegmentationParameters(index_duration=DUR)
    plan = engine.plan(config.to_native(), V, T, C, U, texts_of_31_labels=True)
    assert plan.info["vocab_pitch"] == 34
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    d_lpz = torch.from_numpy(np.concatenate([s[0].reshape(-1) for s in segs])).to(dev)
    d_lab = torch.from_numpy(np.concatenate([s[1] for s in segs]).astype(np.int32)).to(dev)
    d_ub = torch.from_numpy(np.concatenate([s[2] for s in segs]).astype(np.int32)).to(dev)
    o = dict(fol=torch.full((sum(C),), 9, dtype=torch.int32, device=dev), cp=torch.ones(sum(T), dtype=torch.float32, device=dev),
             seg=torch.ones(3, sum(U), dtype=torch.float64, device=dev), te=torch.zeros(B, dtype=torch.int32, device=dev),
             status=torch.full((B,), -7, dtype=torch.int32, device=dev))
    plan.run_device(d_lpz.data_ptr(), d_lab.data_ptr(), d_ub.data_ptr(), o["fol"].data_ptr(), o["cp"].data_ptr(), None,
                    o["seg"][0].data_ptr(), o["seg"][1].data_ptr(), o["seg"][2].data_ptr(), o["te"].data_ptr(),
                    o["status"].data_ptr(), stream)
    torch.cuda.synchronize()
    status = o["status"].cpu().numpy()
    assert status.tolist() == [0, pkg._native.ST_TOO_MANY_LABELS, 0, pkg._native.ST_TOO_MANY_LABELS, 0]
    ocfg = oracle.make_config(index_duration=DUR)
    co = np.concatenate([[0], np.cumsum(C)])
    fol, te = o["fol"].cpu().numpy(), o["te"].cpu().numpy()
    for b, sg in enumerate(segs):
        if status[b] == 0:
            r = oracle.get_segments(*sg, ocfg)
            assert te[b] == r["t_end"] and np.array_equal(fol[co[b]:co[b + 1]], r["frame_of_label"]), b
        else:
            assert te[b] == -1 and not fol[co[b]:co[b + 1]].any()
    plan.close()
    # the host-buffer entry looks at the labels itself: the same batch is simply not narrowed
    _check(pkg, oracle, segs, _run(pkg, segs))


def test_nonzero_blank_index(pkg, oracle):
    syn = pkg.synthetic
    segs = [syn.make_segment(300 + s, 300, 32, 3, 20, blank=31) for s in range(4)]
    _check(pkg, oracle, segs, _run(pkg, segs, blank=31), dict(blank=31))


def test_audio_shorter_than_text_status_and_exception(pkg, oracle):
    syn = pkg.synthetic
    short = syn.make_segment(1, 30, 32, 2, 20)   # C = 44 > T = 30
    ok = syn.make_segment(2, 200, 32, 2, 20)
    res = _run(pkg, [short, ok, short])
    assert [r["status"] for r in res] == [1, 0, 1]
    _check(pkg, oracle, [short, ok, short], res)
    config = pkg.CtcSegmentationParameters(index_duration=DUR)
    with pytest.raises(AssertionError, match="Audio is shorter than text!"):
        pkg.ctc_segmentation.ctc_segmentation(config, short[0], short[1].reshape(-1, 1))


def test_forced_diagonal_T_equals_C(pkg, oracle):
    """T == C leaves exactly one path: every frame switches (known answer)."""
    syn = pkg.synthetic
    lpz, gt, ub = syn.make_segment(5, 62, 32, 2, 29)  # C = 1 + 2*30 + 1 = 62
    assert len(gt) == 62
    res = _run(pkg, [(lpz, gt, ub)])
    assert res[0]["status"] == 0
    assert np.array_equal(res[0]["frame_of_label"], np.arange(62))
    _check(pkg, oracle, [(lpz, gt, ub)], res)


def test_one_hot_emissions_unique_path(pkg, oracle):
    """Near one-hot emissions along a planted path: the alignment must recover it."""
    rng = np.random.default_rng(11)
    V, T = 32, 500
    gt, ub = pkg.synthetic.make_labels(rng, 4, 20, V)
    C = len(gt)
    firsts = np.sort(rng.choice(np.arange(1, T), size=C - 1, replace=False))
    logits = np.full((T, V), -30.0, np.float32)
    col = np.zeros(T, np.int64)
    col[firsts] = 1
    col = np.cumsum(col)
    logits[np.arange(T), 0] = 0.0           # blank everywhere ...
    logits[firsts, :] = -30.0
    logits[firsts, gt[1:]] = 0.0            # ... except the first frame of each label
    lpz = (logits - np.log(np.exp(logits.astype(np.float64)).sum(1, keepdims=True))).astype(np.float32)
    res = _run(pkg, [(lpz, gt, ub)])
    assert res[0]["status"] == 0
    # separator (blank) columns can enter at any blank frame; real labels are pinned
    real = np.flatnonzero(gt[1:] != 0)
    assert np.array_equal(res[0]["frame_of_label"][1:][real], firsts[real])
    _check(pkg, oracle, [(lpz, gt, ub)], res)


@pytest.mark.parametrize("V", [20, 32, 38, 64])
def test_bursts_of_switches_across_block_boundaries(pkg, oracle, V):
    """Planted paths that switch in every frame for 28..40 frames in a row, at all phases relative to
    the 32-row blocks: in checkpoint mode (V <= 64) the backtrack recomputes a block's decisions in a
    64-column window that starts where the path entered the block BEFORE; 32 switches in one block
    push the path out of that window and the block is recomputed from its exact entry column."""
    rng = np.random.default_rng(77 + V + FUZZ_SALT)
    segs = []
    for s in range(6):
        gt, ub = pkg.synthetic.make_labels(rng, 10, 30, V)
        C = len(gt)
        T = 3 * C + int(rng.integers(0, 40))
        firsts, t = [], 1 + int(rng.integers(0, 32))
        while len(firsts) < C - 1:
            run = min(int(rng.integers(28, 41)), C - 1 - len(firsts))
            firsts.extend(range(t, t + run))
            t += run + int(rng.integers(1, 70))
        firsts = np.asarray(firsts)
        assert firsts[-1] < T
        logits = np.full((T, V), -30.0, np.float32) + rng.normal(0, 0.5, (T, V)).astype(np.float32)
        logits[:, 0] += 30.0                     # blank everywhere ...
        logits[firsts, 0] -= 30.0
        logits[firsts, gt[1:]] += 30.0           # ... except the first frame of each label
        lpz = (logits - np.log(np.exp(logits.astype(np.float64)).sum(1, keepdims=True))).astype(np.float32)
        segs.append((lpz, gt, ub))
    for kw in (dict(), dict(backtrack_from_max_t=True), dict(preamble_transition_cost_zero=False)):
        res = _run(pkg, segs, **kw)
        _check(pkg, oracle, segs, res, cfg_kw={k: int(v) for k, v in kw.items()})
        assert all(r["status"] == 0 for r in res)


def test_checkpoint_mode_and_decision_word_mode_agree(pkg, oracle, monkeypatch):
    """V <= 64 runs in checkpoint mode (the fill stores table rows, the backtrack recomputes its
    decisions); CTCFA_DECISION_BITS=1 forces the decision-word mode that wider vocabularies use."""
    syn = pkg.synthetic
    rng = np.random.default_rng(9 + FUZZ_SALT)
    segs = [syn.make_segment(900 + s, int(rng.integers(40, 1500)), 32, int(rng.integers(1, 9)), int(rng.integers(3, 40)))
            for s in range(24)]
    segs = [s for s in segs if len(s[1]) <= s[0].shape[0]]
    monkeypatch.setenv("CTCFA_CHECKPOINT", "1")
    a = _run(pkg, segs)
    monkeypatch.setenv("CTCFA_DECISION_BITS", "1")
    b = _run(pkg, segs)
    monkeypatch.delenv("CTCFA_DECISION_BITS")
    _check(pkg, oracle, segs, a)
    for x, y in zip(a, b):
        assert x["status"] == y["status"] and x["t_end"] == y["t_end"]
        for k in ("frame_of_label", "char_prob", "state", "seg_start", "seg_end", "seg_score"):
            assert np.array_equal(x[k], y[k]), k


def test_benchmark_shaped_segments_match_oracle(pkg, oracle):
    """BASELINE.json configs[2] geometry (3000 frames x 640 label columns), every segment checked
    against the oracle -- in decision-word mode (what the host-buffer entry picks for 8 segments) and,
    through the dp_mode fixture, in checkpoint mode (what the benchmark's 512-segment plan runs in)."""
    syn = pkg.synthetic
    segs = [syn.make_segment(7000 + FUZZ_SALT * 100 + s, 3000, 32, 22, 28) for s in range(8)]
    for kw in (dict(), dict(backtrack_from_max_t=True), dict(preamble_transition_cost_zero=False)):
        res = _run(pkg, segs, **kw)
        _check(pkg, oracle, segs, res, cfg_kw={k: int(v) for k, v in kw.items()})
        assert all(r["status"] == 0 for r in res)


def test_backtrack_from_max_t(pkg, oracle):
    syn = pkg.synthetic
    segs = [syn.make_segment(400 + s, 350, 32, 3, 22) for s in range(3)]
    _check(pkg, oracle, segs, _run(pkg, segs, backtrack_from_max_t=True), dict(backtrack_from_max_t=1))


def test_preamble_cost_not_zero(pkg, oracle):
    syn = pkg.synthetic
    segs = [syn.make_segment(500 + s, 280, 32, 3, 18) for s in range(3)]
    _check(pkg, oracle, segs, _run(pkg, segs, preamble_transition_cost_zero=False),
           dict(preamble_transition_cost_zero=0))


def test_scoring_lengths(pkg, oracle):
    syn = pkg.synthetic
    segs = [syn.make_segment(600 + s, 500, 32, 4, 25) for s in range(2)]
    for L in (1, 7, 8, 30, 64, 128):
        _check(pkg, oracle, segs, _run(pkg, segs, score_min_mean_over_L=L), dict(score_min_mean_over_L=L))


def test_scoring_lengths_above_128_frames(pkg, oracle):
    """score_min_mean_over_L beyond the 128 frames the backtrack kernels sum in one go (rescore_kernel: NumPy's
    pairwise summation at any length) -- utterances longer than L (sliding means) and shorter (one mean over the
    whole span), a failed segment in the batch, and a segment of the windowed regime."""
    syn = pkg.synthetic
    segs = [syn.make_segment(640 + s, T, 32, U, n) for s, (T, U, n) in
            enumerate([(3000, 2, 40), (2200, 5, 20), (1400, 1, 60), (900, 3, 10), (40, 4, 25)])]
    for L in (129, 136, 200, 257, 500, 1023, 3000):
        _check(pkg, oracle, segs, _run(pkg, segs, score_min_mean_over_L=L), dict(score_min_mean_over_L=L))
    kw = dict(min_window_size=1000, max_window_size=8000, score_min_mean_over_L=300)
    _check(pkg, oracle, segs, _run(pkg, segs, **kw), cfg_kw=kw)


def test_longer_segments_full_size_column_count(pkg, oracle):
    """T = 3000, C = 640 (BASELINE.json configs[2] geometry), a handful of segments."""
    syn = pkg.synthetic
    segs = [syn.make_segment(700 + s, 3000, 32, 22, 28) for s in range(6)]
    _check(pkg, oracle, segs, _run(pkg, segs))


def test_sharp_emissions_use_all_frames(pkg, oracle):
    """Strong planted peaks: the best path runs to the last frames (long backtrack)."""
    segs = []
    for s in range(4):
        gt, ub = pkg.synthetic.make_labels(np.random.default_rng(900 + s), 10, 24, 32)
        lpz = pkg.synthetic.make_emissions(np.random.default_rng(800 + s), 1500, 32, gt, noise=1.0, peak=14.0)
        segs.append((lpz, gt, ub))
    res = _run(pkg, segs)
    assert all(r["t_end"] > 1300 for r in res)
    _check(pkg, oracle, segs, res)


def test_speechbrain_protocol_str_task(pkg, oracle):
    """get_lpz -> prepare_segmentation_task -> get_segments -> task.set -> str(task),
    the exact call sequence of iterative_utterance_alignment.py:201-219."""
    from tests.fakes import FakeASR

    asr = FakeASR(seed=3)
    aligner = pkg.CTCSegmentation(asr, kaldi_style_text=False, time_stamps="fixed", scoring_length=30)
    ratio = aligner.estimate_samples_to_frames_ratio()
    assert abs(ratio - 320.0) < 1.0
    import torch
    audio = torch.from_numpy(np.random.default_rng(0).standard_normal(16000 * 6).astype(np.float32))
    lpz = aligner.get_lpz(audio)
    text = ["HOLA QUE TAL", "MUY BIEN GRACIAS", "ADIOS"]
    task = aligner.prepare_segmentation_task(text, lpz, "utt_7", audio.shape[0])
    segments = aligner.get_segments(task)
    task.set(**segments)
    lines = str(task).strip().split("\n")
    assert len(lines) == 3
    fields = [ln.split(" ", 5) for ln in lines]
    assert all(len(f) == 6 for f in fields)
    assert [f[0] for f in fields] == ["utt_7_0000", "utt_7_0001", "utt_7_0002"]
    assert [f[5] for f in fields] == text
    # against the oracle on the same label matrix
    ocfg = oracle.make_config(index_duration=aligner.config.index_duration)
    o = oracle.get_segments(lpz, task.ground_truth_mat[:, 0], task.utt_begin_indices, ocfg)
    for f, s, e, sc in zip(fields, o["seg_start"], o["seg_end"], o["seg_score"]):
        assert f[2] == f"{s:.2f}" and f[3] == f"{e:.2f}"
        assert abs(float(f[4]) - sc) <= SCORE_TOL + 5e-5


def test_golden_vectors(pkg):
    """HIP path against the committed fixtures (tests/golden/dp_vectors.npz; provenance:
    restatement -- see tests/golden/make_dp_goldens.py).  No oracle in the loop."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "dp_vectors.npz"))
    for name in g["names"]:
        blank, pre, maxt, L = (int(x) for x in g[f"{name}/cfg"])
        config = pkg.CtcSegmentationParameters(index_duration=float(g["index_duration"]), blank=blank,
                                               preamble_transition_cost_zero=bool(pre),
                                               backtrack_from_max_t=bool(maxt), score_min_mean_over_L=L)
        r = pkg.ctc_segmentation.get_segments_device(config, [g[f"{name}/lpz"]], [g[f"{name}/gt"]],
                                                     [g[f"{name}/utt_begin"]])[0]
        assert r["status"] == int(g[f"{name}/status"]), name
        if r["status"] != 0:
            continue
        assert r["t_end"] == int(g[f"{name}/t_end"]), name
        assert np.array_equal(r["frame_of_label"], g[f"{name}/frame_of_label"]), name
        assert np.array_equal(r["char_prob"], g[f"{name}/char_probs"]), name
        assert np.array_equal(r["state"], g[f"{name}/state"]), name
        assert np.array_equal(r["seg_start"], g[f"{name}/seg_start"]) and np.array_equal(r["seg_end"], g[f"{name}/seg_end"])
        np.testing.assert_allclose(r["seg_score"], g[f"{name}/seg_score"], rtol=0, atol=SCORE_TOL)


def test_full_size_properties(pkg, engine):
    """BASELINE.json configs[2] at full size (512 x 3000 x 32, C = 640) through the plan API:
    size-independent properties instead of an oracle run -- monotone label frames, every frame
    in (0, t_end] visited exactly once, scores <= 0, and batch-position independence
    (the same segment placed at two batch positions gives identical results)."""
    import torch
    syn = pkg.synthetic
    B, T, V, U, n = 512, 3000, 32, 22, 28
    base = [syn.make_segment(s, T, V, U, n) for s in range(8)]
    idx = np.arange(B) % 8
    lpz = np.stack([base[i][0] for i in idx])
    gt = np.stack([base[i][1] for i in idx])
    ub = np.stack([base[i][2] for i in idx])
    C = gt.shape[1]
    config = pkg.CtcSegmentationParameters(index_duration=DUR)
    plan = engine.plan(config.to_native(), V, [T] * B, [C] * B, [U] * B)
    dev = torch.device("cuda:0")
    d = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a.reshape(-1)).astype(dt)).to(dev)
    d_lpz, d_lab, d_ub = d(lpz, np.float32), d(gt, np.int32), d(ub, np.int32)
    fol = torch.empty(B * C, dtype=torch.int32, device=dev)
    cp = torch.empty(B * T, dtype=torch.float32, device=dev)
    st = torch.empty(B * T, dtype=torch.int32, device=dev)
    seg = torch.empty(3, B * U, dtype=torch.float64, device=dev)
    te = torch.empty(B, dtype=torch.int32, device=dev)
    status = torch.empty(B, dtype=torch.int32, device=dev)
    plan.run_device(d_lpz.data_ptr(), d_lab.data_ptr(), d_ub.data_ptr(), fol.data_ptr(), cp.data_ptr(), st.data_ptr(),
                    seg[0].data_ptr(), seg[1].data_ptr(), seg[2].data_ptr(), te.data_ptr(), status.data_ptr(),
                    torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    fol, cp, st = fol.cpu().numpy().reshape(B, C), cp.cpu().numpy().reshape(B, T), st.cpu().numpy().reshape(B, T)
    seg, te, status = seg.cpu().numpy().reshape(3, B, U), te.cpu().numpy(), status.cpu().numpy()
    assert (status == 0).all()
    assert (fol[:, 0] == 0).all() and (np.diff(fol[:, 1:], axis=1) > 0).all()
    t = np.arange(T)[None, :]
    visited = st != -2
    assert np.array_equal(visited, (t >= 1) & (t <= te[:, None]))
    assert (cp <= 0).all() and (seg[2] <= 0).all() and (seg[1] >= seg[0]).all()
    for b in range(8, B):  # replicas of the 8 base segments must agree bit for bit
        assert np.array_equal(fol[b], fol[b % 8]) and np.array_equal(cp[b], cp[b % 8])
        assert np.array_equal(seg[:, b], seg[:, b % 8])
    plan.close()


@pytest.mark.parametrize("min_window", [8000, 256])
def test_pipelined_runs_match_serial(pkg, engine, min_window):
    """ctcfa_plan_run_pipelined (backtrack on the plan's side stream, two workspaces) must give
    the results of the serial entry for every call in a sequence.  min_window 256 sends every
    segment (T = 700) through the windowed kernel, which shares ONE table workspace between runs."""
    import torch
    syn = pkg.synthetic
    batches = [[syn.make_segment(50 * k + s, 700, 32, 5, 24) for s in range(16)] for k in range(4)]
    config = pkg.CtcSegmentationParameters(index_duration=DUR, min_window_size=min_window)
    T, C, U = [700] * 16, [len(batches[0][0][1])] * 16, [5] * 16
    plan = engine.plan(config.to_native(), 32, T, C, U)
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream

    def run(batch, pipelined):
        d_lpz = torch.from_numpy(np.concatenate([s[0].reshape(-1) for s in batch])).to(dev)
        d_lab = torch.from_numpy(np.concatenate([s[1] for s in batch]).astype(np.int32)).to(dev)
        d_ub = torch.from_numpy(np.concatenate([s[2] for s in batch]).astype(np.int32)).to(dev)
        o = dict(fol=torch.zeros(sum(C), dtype=torch.int32, device=dev), cp=torch.zeros(sum(T), dtype=torch.float32, device=dev),
                 seg=torch.zeros(3, sum(U), dtype=torch.float64, device=dev), te=torch.zeros(16, dtype=torch.int32, device=dev),
                 status=torch.zeros(16, dtype=torch.int32, device=dev), keep=(d_lpz, d_lab, d_ub))
        plan.run_device(d_lpz.data_ptr(), d_lab.data_ptr(), d_ub.data_ptr(), o["fol"].data_ptr(), o["cp"].data_ptr(), None,
                        o["seg"][0].data_ptr(), o["seg"][1].data_ptr(), o["seg"][2].data_ptr(), o["te"].data_ptr(),
                        o["status"].data_ptr(), stream, pipelined=pipelined)
        return o

    serial = []
    for b in batches:
        o = run(b, False)
        torch.cuda.synchronize()
        serial.append({k: v.cpu().numpy() for k, v in o.items() if k != "keep"})
    piped = [run(b, True) for b in batches]
    plan.flush(stream)
    torch.cuda.synchronize()
    for s, o in zip(serial, piped):
        for k in ("fol", "cp", "seg", "te", "status"):
            assert np.array_equal(s[k], o[k].cpu().numpy()), k
    plan.close()


def _quantized_segment(seed, T, V, U, n, step, lo=-12.0):
    """Emissions on a coarse grid: a + e and b + m collide exactly all the time, so the package's
    residual tie rule (strict '>': ties stay) and fp32 rounding decide most transitions."""
    rng = np.random.default_rng(seed)
    gt, ub = __import__("importlib").import_module("iterative-pseudo-forced-alignment-ctc_amd.synthetic").make_labels(rng, U, n, V)
    lpz = (np.round(rng.uniform(lo, 0.0, size=(T, V)) / step) * step).astype(np.float32)
    return lpz, gt, ub


@pytest.mark.parametrize("step", [0.0, 0.5, 0.125, 1.0 / 3.0])
def test_exact_ties_and_rounding(pkg, oracle, step):
    """step == 0: perfectly uniform emissions (every path ties); otherwise a coarse grid."""
    segs = []
    for s in range(6):
        T, U, n = 200 + 37 * s, 2 + s % 3, 10 + 3 * s
        if step == 0.0:
            gt, ub = pkg.synthetic.make_labels(np.random.default_rng(s), U, n, 32)
            lpz = np.full((T, 32), np.log(1.0 / 32.0), np.float32)
        else:
            lpz, gt, ub = _quantized_segment(1300 + s, T, 32, U, n, step)
        segs.append((lpz, gt, ub))
    _check(pkg, oracle, segs, _run(pkg, segs))


def test_large_magnitude_emissions(pkg, oracle):
    """Scores reach -1e5: ulp(table) ~ 0.008, so rounding noise decides many near-ties."""
    segs = []
    for s in range(4):
        rng = np.random.default_rng(1400 + s)
        gt, ub = pkg.synthetic.make_labels(rng, 4, 20, 32)
        lpz = rng.uniform(-90.0, -1.0, size=(1200, 32)).astype(np.float32)
        segs.append((lpz, gt, ub))
    _check(pkg, oracle, segs, _run(pkg, segs))


def test_many_random_small_cases(pkg, oracle):
    """300 random tiny segments in one launch (C from 2 to ~70, T barely above C included)."""
    rng = np.random.default_rng(99)
    segs = []
    for s in range(300):
        U = int(rng.integers(0, 4))
        n = int(rng.integers(1, 16))
        C = 1 + U * (1 + n) + 1 if U else 2
        T = int(C + rng.integers(0, 40))
        V = 32
        gt, ub = pkg.synthetic.make_labels(np.random.default_rng(2000 + s), U, n, V)
        kind = s % 3
        if kind == 0:
            lpz = pkg.synthetic.make_emissions(np.random.default_rng(3000 + s), T, V, gt)
        elif kind == 1:
            lpz = (np.round(rng.uniform(-8, 0, size=(T, V)) * 4) / 4).astype(np.float32)
        else:
            lpz = pkg.synthetic.make_emissions(np.random.default_rng(3000 + s), T, V, gt, noise=0.5, peak=15.0)
        segs.append((lpz, gt, ub))
    res = _run(pkg, segs)
    assert sum(r["status"] == 0 for r in res) == len(segs)
    _check(pkg, oracle, segs, res)


@pytest.mark.parametrize("where", ["everywhere", "late", "first_row", "one_value"])
def test_emissions_that_are_not_log_probabilities(pkg, oracle, where):
    """The fill kernel idles through trellis regions that hold exactly -1e9 while every
    emission so far is <= 0; positive inputs (raw logits) must switch that off -- from the
    first block, or in the middle of a segment -- and still match the oracle bit for bit."""
    segs = []
    for s in range(4):
        rng = np.random.default_rng(1500 + s)
        T, U, n = 900 + 101 * s, 10 + s, 24
        gt, ub = pkg.synthetic.make_labels(rng, U, n, 32)
        lpz = pkg.synthetic.make_emissions(rng, T, 32, gt)
        if where == "everywhere":
            lpz = (lpz + 7.5).astype(np.float32)
        elif where == "late":
            lpz[T // 3:] += np.float32(9.0)
        elif where == "first_row":
            lpz[:2] += np.float32(11.0)
        else:
            lpz[40 + 13 * s, 5] = np.float32(3.25)
        segs.append((lpz, gt, ub))
    _check(pkg, oracle, segs, _run(pkg, segs))


# ---- windowed regime: T > min_window_size -------------------------------------
def _windowed_cases():
    cases = []
    for s, (T, U, n, mw) in enumerate([(400, 6, 12, 80), (700, 9, 15, 128), (1000, 12, 14, 100),
                                       (333, 3, 20, 64), (900, 20, 9, 256), (520, 4, 30, 500)]):
        cases.append((1700 + s, T, U, n, mw))
    return cases


@pytest.mark.parametrize("max_window", [1000, 300])
def test_windowed_regime_matches_oracle(pkg, oracle, max_window):
    """Small min_window_size so that the per-column window offsets, window doubling after a
    failed backtrack and the final IndexError (max_window=300) all occur at test sizes."""
    seen_status = set()
    for seed, T, U, n, mw in _windowed_cases():
        seg = pkg.synthetic.make_segment(seed, T, 32, U, n)
        kw = dict(min_window_size=mw, max_window_size=max_window)
        res = _run(pkg, [seg], **kw)
        _check(pkg, oracle, [seg], res, cfg_kw=kw)
        seen_status.add(int(res[0]["status"]))
    assert 0 in seen_status


def test_windowed_and_plain_segments_in_one_batch(pkg, oracle):
    """A batch that mixes both regimes: T <= min_window_size goes through fill + backtrack,
    T > min_window_size through the windowed kernel; results land in the same arrays."""
    mw = 300
    segs = [pkg.synthetic.make_segment(1800 + s, T, 32, U, n)
            for s, (T, U, n) in enumerate([(250, 4, 12), (900, 10, 14), (300, 5, 10), (301, 5, 10), (1500, 12, 20)])]
    kw = dict(min_window_size=mw, max_window_size=4000)
    _check(pkg, oracle, segs, _run(pkg, segs, **kw), cfg_kw=kw)


def test_windowed_regime_noisy_and_flat_emissions(pkg, oracle):
    """Emissions without a planted path: the window offsets follow the noise, backtracks fail
    and windows double more often."""
    segs = []
    for s in range(4):
        rng = np.random.default_rng(1900 + s)
        gt, ub = pkg.synthetic.make_labels(rng, 5 + s, 10, 32)
        T = 500 + 120 * s
        lpz = np.log(rng.dirichlet(np.ones(32) * (0.3 if s % 2 else 5.0), size=T)).astype(np.float32)
        segs.append((lpz, gt, ub))
    for mw, mx in ((64, 2000), (100, 400)):
        kw = dict(min_window_size=mw, max_window_size=mx)
        _check(pkg, oracle, segs, _run(pkg, segs, **kw), cfg_kw=kw)


@pytest.mark.parametrize("paths", ["rows+wave", "rows+literal", "literal+wave", "literal+literal"])
def test_windowed_regime_fill_and_walk_paths_agree(pkg, oracle, monkeypatch, paths):
    """The windowed regime has two fills (band_fill_kernel: row by row under guessed offsets that the result proves; the
    literal column-by-column one) and two walks (64 steps at a time on plain cells; the literal step): every combination
    gives the oracle's results -- planted paths, noise, flat rows, vocabularies that are not 32 entries, every flag, texts
    of more than 1 024 / 2 048 label columns (two / four columns per lane), emissions that hold NaN and infinities."""
    fill, walk = paths.split("+")
    if fill == "literal":
        monkeypatch.setenv("CTCFA_NO_BAND_FILL", "1")
    if walk == "literal":
        monkeypatch.setenv("CTCFA_NO_FAST_WALK", "1")
    syn = pkg.synthetic
    for seed, T, U, n, mw in _windowed_cases():
        seg = syn.make_segment(seed, T, 32, U, n)
        kw = dict(min_window_size=mw, max_window_size=1000)
        _check(pkg, oracle, [seg], _run(pkg, [seg], **kw), cfg_kw=kw)
    # noise and flat rows, other vocabularies, the flags
    segs = []
    for s in range(6):
        rng = np.random.default_rng(2100 + s)
        V = (32, 29, 38, 76, 200, 5)[s]
        gt, ub = syn.make_labels(rng, 5 + s, 10, V)
        T = 500 + 120 * s
        lpz = (np.log(rng.dirichlet(np.ones(V) * (0.3 if s % 2 else 5.0), size=T)).astype(np.float32) if s < 4
               else syn.make_emissions(rng, T, V, gt))
        segs.append((lpz, gt, ub, V))
    for lpz, gt, ub, V in segs:
        for kw in (dict(min_window_size=64, max_window_size=4000), dict(min_window_size=150, max_window_size=4000, backtrack_from_max_t=True),
                   dict(min_window_size=100, max_window_size=4000, preamble_transition_cost_zero=False)):
            _check(pkg, oracle, [(lpz, gt, ub)], _run(pkg, [(lpz, gt, ub)], **kw), cfg_kw=kw)
    # long texts: more than one column per lane of the row-by-row fill
    for seed, T, U, n, mw in ((2200, 2600, 60, 20, 1500), (2201, 4500, 110, 20, 3000)):
        seg = syn.make_segment(seed, T, 32, U, n)
        kw = dict(min_window_size=mw, max_window_size=100000)
        _check(pkg, oracle, [seg], _run(pkg, [seg], **kw), cfg_kw=kw)
    # emissions that are not numbers
    rng = np.random.default_rng(2300)
    for k in range(6):
        lpz, gt, ub = syn.make_segment(2300 + k, 700, 32, 8, 12)
        lpz = lpz.copy()
        rows = rng.integers(1, 700, size=5)
        cols = rng.integers(0, 32, size=5)
        lpz[rows, cols] = (np.nan, np.inf, -np.inf, np.nan, 3.0e38)[k % 5]
        if k == 5:
            lpz[rng.integers(1, 700, size=3), 0] = np.nan   # the blank itself
        kw = dict(min_window_size=200, max_window_size=3000)
        r = _run(pkg, [(lpz, gt, ub)], **kw)[0]
        o = oracle.get_segments(lpz, gt, ub, oracle.make_config(index_duration=DUR, **kw))
        assert r["status"] == o["status"], k
        if o["status"] == 0:
            assert r["t_end"] == o["t_end"], k
            assert np.array_equal(r["frame_of_label"], o["frame_of_label"]), k
            assert np.array_equal(r["char_prob"].astype(np.float64), o["char_probs"], equal_nan=True), k
            assert np.array_equal(r["state"], o["state"]), k


def test_windowed_regime_production_window(pkg, oracle):
    """The package defaults (min_window_size=8000): a 190 s window, T = 9500 frames."""
    seg = pkg.synthetic.make_segment(1950, 9500, 32, 40, 30)
    _check(pkg, oracle, [seg], _run(pkg, [seg]))


@pytest.mark.parametrize("mw", [8, 12, 16, 24])
def test_windowed_regime_failed_backtracks_and_window_doubling(pkg, oracle, mw):
    """Peaky random emissions and tiny windows: about one backtrack in five leaves the table
    (the package's IndexError), the window doubles once or twice, or -- with max_window_size
    just above the first window -- the error is final (status 2).  ~100 segments per launch."""
    segs = []
    for seed in range(4000, 4400):
        rng = np.random.default_rng(seed)
        U = int(rng.integers(2, 10))
        n = int(rng.integers(4, 12))
        gt, ub = pkg.synthetic.make_labels(rng, U, n, 32)
        T = int(len(gt) * rng.uniform(1.2, 8))
        w = int(rng.choice([8, 12, 16, 24]))
        alpha = float(rng.choice([0.02, 0.05, 0.1]))
        if w != mw or T <= w:
            continue
        lpz = np.log(rng.dirichlet(np.ones(32) * alpha, size=T) + 1e-30).astype(np.float32)
        segs.append((lpz, gt, ub))
    assert len(segs) > 50
    n_fail = {}
    for mx in (mw + 1, 2 * mw + 1, 100000):
        kw = dict(min_window_size=mw, max_window_size=mx)
        res = _run(pkg, segs, **kw)
        _check(pkg, oracle, segs, res, cfg_kw=kw)
        n_fail[mx] = sum(int(r["status"]) == 2 for r in res)
    assert n_fail[mw + 1] >= 3 and n_fail[100000] == 0 and n_fail[2 * mw + 1] < n_fail[mw + 1]


def _fuzz_segment(rng, V, blank, Tmax):
    U = int(rng.integers(0, 9))
    n = int(rng.integers(1, 30))
    gt, ub = __import__("importlib").import_module("iterative-pseudo-forced-alignment-ctc_amd").synthetic.make_labels(rng, U, n, V, blank=blank)
    C = len(gt)
    lo = max(2, C - 3)                      # a few segments with T < C (status 1)
    T = int(rng.integers(lo, max(lo + 1, min(Tmax, 6 * C + 40))))
    kind = int(rng.integers(0, 5))
    if kind == 0:       # flat: every path ties
        lpz = np.full((T, V), np.float32(np.log(1.0 / V)))
    elif kind == 1:     # coarse grid
        lpz = (np.round(rng.uniform(-12, 0, size=(T, V)) * 2) / 2).astype(np.float32)
    elif kind == 2:     # peaky noise
        lpz = np.log(rng.dirichlet(np.ones(V) * 0.1, size=T) + 1e-30).astype(np.float32)
    elif kind == 3:     # raw logits (positive values: the -1e9 shortcut must switch off)
        lpz = (rng.standard_normal((T, V)) * 4).astype(np.float32)
    else:               # smooth log-softmax noise
        x = rng.standard_normal((T, V)) * 2
        lpz = (x - np.log(np.exp(x).sum(1, keepdims=True))).astype(np.float32)
    return lpz, gt, ub


@pytest.mark.parametrize("V,blank", [(5, 0), (17, 3), (32, 0), (33, 32), (38, 0), (47, 46), (55, 9), (63, 62), (64, 1), (76, 70), (90, 3), (100, 0), (111, 64), (128, 127)])
@pytest.mark.parametrize("flags", [dict(), dict(preamble_transition_cost_zero=False), dict(backtrack_from_max_t=True)])
def test_fuzz_shapes_vocabularies_and_flags(pkg, oracle, V, blank, flags):
    """Random ragged batches over vocabulary sizes (all three LDS row pitches), blank positions,
    flag combinations and scoring lengths; emissions from exact ties to raw logits."""
    rng = np.random.default_rng(7000 + 131 * V + blank + 17 * len(flags) + 1_000_003 * FUZZ_SALT)
    segs = [_fuzz_segment(rng, V, blank, 700) for _ in range(40)]
    L = int(rng.integers(1, 129))
    kw = dict(blank=blank, score_min_mean_over_L=L, **flags)
    res = _run(pkg, segs, **kw)
    _check(pkg, oracle, segs, res, cfg_kw=kw)
    assert any(r["status"] == 0 for r in res)


@pytest.mark.parametrize("batch", [3, 300])
def test_path_along_the_dead_zone_boundary(pkg, oracle, batch):
    """backtrack_from_max_t with flat emissions: the path is the last diagonal that can still
    reach the end cell, i.e. it runs along the edge of the region the fill kernel skips, and with
    T - C == 1 (mod 32) every tile stops exactly at a block end (regression: the tile's last row
    was never handed to the next tile).  batch 300 takes the mixed 8-wave shape."""
    segs = []
    for s in range(3):
        gt, ub = pkg.synthetic.make_labels(np.random.default_rng(5000 + s), 22, 28, 32)
        C = len(gt)
        T = C + 1 + 32 * (3 + 4 * s)
        segs.append((np.full((T, 32), np.float32(np.log(1.0 / 32.0))), gt, ub))
    segs = [segs[i % 3] for i in range(batch)]
    for kw in (dict(backtrack_from_max_t=True), dict()):
        res = _run(pkg, segs, **kw)
        _check(pkg, oracle, segs[:3], res[:3], cfg_kw=kw)
        for i in range(3, batch):
            assert np.array_equal(res[i]["frame_of_label"], res[i % 3]["frame_of_label"])


def _run_plan(pkg, engine, segs, K=0, **cfg):
    """Plan API with device buffers and an optional forced tile width; same result dicts as _run."""
    import torch

    config = pkg.CtcSegmentationParameters(index_duration=DUR, **cfg)
    V = segs[0][0].shape[1]
    T = [s[0].shape[0] for s in segs]
    C = [len(s[1]) for s in segs]
    U = [len(s[2]) - 1 for s in segs]
    plan = engine.plan(config.to_native(), V, T, C, U, force_cols_per_lane=K)
    dev = torch.device("cuda:0")
    d_lpz = torch.from_numpy(np.concatenate([s[0].reshape(-1) for s in segs])).to(dev)
    d_lab = torch.from_numpy(np.concatenate([s[1] for s in segs]).astype(np.int32)).to(dev)
    d_ub = torch.from_numpy(np.concatenate([s[2] for s in segs]).astype(np.int32)).to(dev)
    nT, nC, nU = sum(T), sum(C), max(1, sum(U))
    d_fol = torch.empty(nC, dtype=torch.int32, device=dev)
    d_cp = torch.empty(nT, dtype=torch.float32, device=dev)
    d_st = torch.empty(nT, dtype=torch.int32, device=dev)
    d_seg = torch.empty(3, nU, dtype=torch.float64, device=dev)
    d_te = torch.empty(len(segs), dtype=torch.int32, device=dev)
    d_status = torch.empty(len(segs), dtype=torch.int32, device=dev)
    plan.run_device(d_lpz.data_ptr(), d_lab.data_ptr(), d_ub.data_ptr(), d_fol.data_ptr(), d_cp.data_ptr(),
                    d_st.data_ptr(), d_seg[0].data_ptr(), d_seg[1].data_ptr(), d_seg[2].data_ptr(),
                    d_te.data_ptr(), d_status.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    info = plan.info
    plan.close()
    fol, cp, st = d_fol.cpu().numpy(), d_cp.cpu().numpy(), d_st.cpu().numpy()
    seg, te, status = d_seg.cpu().numpy(), d_te.cpu().numpy(), d_status.cpu().numpy()
    to, co, uo = (np.concatenate([[0], np.cumsum(x)]) for x in (T, C, U))
    return [dict(status=int(status[b]), t_end=int(te[b]), frame_of_label=fol[co[b]:co[b + 1]],
                 char_prob=cp[to[b]:to[b + 1]], state=st[to[b]:to[b + 1]], seg_start=seg[0][uo[b]:uo[b + 1]],
                 seg_end=seg[1][uo[b]:uo[b + 1]], seg_score=seg[2][uo[b]:uo[b + 1]]) for b in range(len(segs))], info


@pytest.mark.parametrize("seed", range(8))
def test_fuzz_medium_shapes_tiles_and_zones(pkg, oracle, engine, seed):
    """Ragged batches up to T = 2500 / C = 900 over forced tile widths (and the automatic choice,
    which takes the mixed 8-wave shape for the large batches), with paths that end anywhere from
    the first feasible frame to T-1: the regions the fill kernel skips change with every
    segment.  Flags and scoring length vary per run."""
    rng = np.random.default_rng(8000 + seed + 1_000_003 * FUZZ_SALT)
    syn = pkg.synthetic
    K = [0, 1, 2, 3, 5, 0, 2, 4][seed]
    batch = 270 if seed in (0, 5) else 20
    segs = []
    for s in range(batch):
        U = int(rng.integers(1, 30))
        n = int(rng.integers(2, 30))
        gt, ub = syn.make_labels(rng, U, n, 32)
        C = len(gt)
        T = int(C + rng.integers(0, 4)) if rng.random() < 0.2 else int(rng.integers(C, max(C + 1, 2500)))
        kind = int(rng.integers(0, 4))
        if kind == 0:
            lpz = np.full((T, 32), np.float32(np.log(1.0 / 32)))
        elif kind == 1:   # sharp: the best path uses every frame
            lpz = syn.make_emissions(rng, T, 32, gt, noise=0.5, peak=15.0)
        elif kind == 2:
            lpz = syn.make_emissions(rng, T, 32, gt)
        else:
            lpz = (np.round(rng.uniform(-9, 0, size=(T, 32)) * 2) / 2).astype(np.float32)
        segs.append((lpz, gt, ub))
    kw = dict(score_min_mean_over_L=int(rng.integers(1, 129)))
    if seed % 3 == 1:
        kw["backtrack_from_max_t"] = True
    if seed % 4 == 2:
        kw["preamble_transition_cost_zero"] = False
    if K:   # a forced tile width covers 15 tiles of (64 - halo lanes) * K columns at most
        cap = 15 * (64 - -(-16 // K)) * K - (K - 1)
        segs = [s for s in segs if len(s[1]) <= cap]
    res, info = _run_plan(pkg, engine, segs, K, **kw)
    if K:
        assert info["cols_per_lane"] == K
    _check(pkg, oracle, segs, res, cfg_kw=kw)


@pytest.mark.parametrize("V,blank", [(5, 0), (32, 7), (64, 63), (128, 0)])
@pytest.mark.parametrize("flags", [dict(), dict(preamble_transition_cost_zero=False), dict(backtrack_from_max_t=True)])
def test_fuzz_windowed_regime(pkg, oracle, V, blank, flags):
    """The windowed kernel over vocabularies, blank positions and flags; window sizes from 8 to
    64 frames with every segment longer than the window, emissions from ties to raw logits."""
    rng = np.random.default_rng(9000 + 7 * V + blank + 3 * len(flags) + 1_000_003 * FUZZ_SALT)
    mw = int(rng.choice([8, 16, 33, 64]))
    segs = []
    while len(segs) < 30:
        s = _fuzz_segment(rng, V, blank, 500)
        if s[0].shape[0] > mw:
            segs.append(s)
    for mx in (2 * mw + 1, 100000):
        kw = dict(blank=blank, min_window_size=mw, max_window_size=mx,
                  score_min_mean_over_L=int(rng.integers(1, 129)), **flags)
        _check(pkg, oracle, segs, _run(pkg, segs, **kw), cfg_kw=kw)


def test_wide_segments_in_a_full_batch(pkg, oracle, engine):
    """C in (1024, 1280] with two workgroups per CU: eight or more tiles per segment, every one of
    them refreshing its halo from its neighbour twice per 32-row block."""
    syn = pkg.synthetic
    base = [syn.make_segment(6000 + s, T, 32, U, n) for s, (T, U, n) in
            enumerate([(2000, 44, 27), (1700, 40, 29), (1500, 41, 30)])]
    assert all(1024 < len(s[1]) <= 1280 for s in base)
    segs = [base[i % 3] for i in range(510)]
    res, info = _run_plan(pkg, engine, segs)
    K, W = info["cols_per_lane"], info["waves_per_seg"]
    assert W * (64 - -(-16 // K)) * K >= max(len(s[1]) for s in base), info   # the tiles cover the widest segment
    _check(pkg, oracle, base, res[:3])
    for i in range(3, len(segs)):
        assert np.array_equal(res[i]["frame_of_label"], res[i % 3]["frame_of_label"])
        assert np.array_equal(res[i]["seg_score"], res[i % 3]["seg_score"])


@pytest.mark.parametrize("remap", [False, True], ids=["gather-kernel", "compact-remap"])
@pytest.mark.parametrize("V,blank", [(257, 0), (500, 0), (1000, 37), (4096, 4095)])
def test_wide_vocabularies(pkg, oracle, monkeypatch, V, blank, remap):
    """V > 256 (sub-word CTC models).  Default: the columns a segment can look at (its labels and the blank, at
    most 256 per emission block) are gathered into a compact matrix and the staged kernels run on renumbered
    labels (`state` comes back in the caller's ids).  CTCFA_NO_REMAP=1, or more than 255 distinct labels: the
    gather kernel -- no LDS staging of vocabulary rows, every lane gathers its own column's emission.
    Ragged batch incl. a segment with T < C and one with C = 2."""
    monkeypatch.setenv("CTCFA_NO_REMAP" if not remap else "CTCFA_REMAP", "1")
    rng = np.random.default_rng(10_000 + V + 1_000_003 * FUZZ_SALT)
    segs = []
    for s in range(12):
        U = int(rng.integers(0, 7)) if s else 0
        n = int(rng.integers(3, 40))
        gt, ub = pkg.synthetic.make_labels(rng, U, n, V, blank=blank)
        C = len(gt)
        T = C - 2 if s == 3 else int(rng.integers(C, 4 * C + 60))
        T = max(T, 2)
        if s % 3 == 0:
            lpz = pkg.synthetic.make_emissions(rng, T, V, gt, blank=blank)
        elif s % 3 == 1:
            x = rng.standard_normal((T, V)) * 2
            lpz = (x - np.log(np.exp(x).sum(1, keepdims=True))).astype(np.float32)
        else:
            lpz = (np.round(rng.uniform(-9, 0, size=(T, V)) * 2) / 2).astype(np.float32)
        segs.append((lpz, gt, ub))
    for kw in (dict(blank=blank), dict(blank=blank, backtrack_from_max_t=True, score_min_mean_over_L=7)):
        _check(pkg, oracle, segs, _run(pkg, segs, **kw), cfg_kw=kw)


@pytest.mark.parametrize("V", [100, 130, 200, 256])
def test_vocabularies_of_65_to_256_entries_take_the_flags_and_long_texts(pkg, oracle, V):
    """Above 64 entries the emission ring holds e alone and the tiles work out m = max(blank, e): every flag but
    gratis_blank (checkpoint mode only), a blank that is not entry 0, texts beyond the gather kernel's 961 columns,
    the windowed regime, shared emissions; one batch wide enough for several tiles with two columns per lane."""
    rng = np.random.default_rng(30_000 + V)
    blank = int(rng.integers(0, V))
    segs = []
    for U, n, T in ((9, 130, 3000), (12, 40, 1300), (3, 12, 150), (1, 2, 9), (5, 30, 100)):
        gt, ub = pkg.synthetic.make_labels(rng, U, n, V, blank=blank)
        segs.append((pkg.synthetic.make_emissions(rng, T, V, gt, blank=blank), gt, ub))
    assert max(len(s[1]) for s in segs) > 961
    for kw in (dict(blank=blank), dict(blank=blank, preamble_transition_cost_zero=False),
               dict(blank=blank, backtrack_from_max_t=True, score_min_mean_over_L=9),
               dict(blank=blank, min_window_size=400, max_window_size=6000)):
        _check(pkg, oracle, segs, _run(pkg, segs, **kw), cfg_kw=kw)
    lpz, gt, ub = segs[1]
    members = [(lpz, gt[:ub[k] + 1].copy(), ub[:k + 1].copy()) for k in (12, 11, 9)]
    config = pkg.CtcSegmentationParameters(index_duration=DUR, blank=blank)
    res = pkg.ctc_segmentation.get_segments_device(config, [lpz] * 3, [m[1] for m in members], [m[2] for m in members])
    _check(pkg, oracle, members, res, cfg_kw=dict(blank=blank))


def test_wide_vocabulary_long_label_sequences(pkg, oracle):
    """Up to 961 label columns (15 waves x 64 lanes) and the windowed regime with V = 300."""
    rng = np.random.default_rng(10_500)
    segs = []
    for U, n, T in ((7, 136, 2500), (20, 40, 1000), (6, 20, 300)):
        gt, ub = pkg.synthetic.make_labels(rng, U, n, 300)
        segs.append((pkg.synthetic.make_emissions(rng, T, 300, gt), gt, ub))
    assert max(len(s[1]) for s in segs) == 961
    _check(pkg, oracle, segs, _run(pkg, segs))
    kw = dict(min_window_size=200, max_window_size=5000)
    _check(pkg, oracle, segs[1:], _run(pkg, segs[1:], **kw), cfg_kw=kw)
    # (the first text has 299 distinct labels: too many for the compact matrix, the batch takes the gather kernel,
    # which is built for the package's default flags only)
    with pytest.raises(NotImplementedError):
        _run(pkg, segs, preamble_transition_cost_zero=False)


def test_blank_transition_cost_zero_between_65_and_256_entries(pkg, oracle):
    """gratis_blank exists in checkpoint mode only, which stages at most 64 vocabulary columns: a vocabulary of
    65..256 entries takes it through the compact matrix when the launch looks at no more than 63 distinct labels."""
    for V in (65, 100, 128, 200, 256):
        rng = np.random.default_rng(21_000 + V)
        blank = int(rng.integers(0, V))
        segs = []
        for s in range(6):
            gt, ub = pkg.synthetic.make_labels(rng, int(rng.integers(1, 4)), int(rng.integers(3, 10)), V, blank=blank)
            T = int(rng.integers(len(gt), 3 * len(gt) + 60))
            segs.append((pkg.synthetic.make_emissions(rng, T, V, gt, blank=blank), gt, ub))
        kw = dict(blank=blank, blank_transition_cost_zero=True)
        _check(pkg, oracle, segs, _run(pkg, segs, **kw), cfg_kw=kw)


@pytest.mark.parametrize("V", [300, 2000])
def test_wide_vocabulary_through_the_compact_matrix_takes_every_flag(pkg, oracle, monkeypatch, V):
    """Texts of at most 127 distinct labels over a wide vocabulary: the staged kernels on the compact matrix --
    with the flags the gather kernel refuses (no preamble_transition_cost_zero, blank_transition_cost_zero when the
    compact vocabulary has at most 64 entries), a blank that is not entry 0, shared emissions, the windowed regime."""
    monkeypatch.setenv("CTCFA_REMAP", "1")   # (also for the default flags and the short texts of this test)
    rng = np.random.default_rng(20_000 + V)
    blank = V - 3
    segs = []
    for s in range(8):
        U = int(rng.integers(1, 5))
        n = int(rng.integers(3, 12))
        gt, ub = pkg.synthetic.make_labels(rng, U, n, V, blank=blank)
        T = int(rng.integers(len(gt), 3 * len(gt) + 80))
        segs.append((pkg.synthetic.make_emissions(rng, T, V, gt, blank=blank), gt, ub))
    for kw in (dict(blank=blank), dict(blank=blank, preamble_transition_cost_zero=False),
               dict(blank=blank, blank_transition_cost_zero=True), dict(blank=blank, min_window_size=64, max_window_size=4000)):
        _check(pkg, oracle, segs, _run(pkg, segs, **kw), cfg_kw=kw)
    # one window, its text and the text minus its last utterance (shared emissions: one compact matrix, one fill)
    lpz, gt, ub = segs[3] if len(segs[3][2]) > 2 else segs[0]
    if len(ub) > 2:
        config = pkg.CtcSegmentationParameters(index_duration=DUR, blank=blank)
        res = pkg.ctc_segmentation.get_segments_device(config, [lpz, lpz], [gt, gt[:ub[-2] + 1]], [ub, ub[:-1]])
        _check(pkg, oracle, [(lpz, gt, ub), (lpz, gt[:ub[-2] + 1], ub[:-1])], res, cfg_kw=dict(blank=blank))


def test_serial_call_after_pipelined_calls_waits_for_the_pending_backtrack(pkg, engine):
    """run_device right after run_pipelined (no flush in between) reuses workspace 0: it has to
    wait for the side-stream backtrack that may still be reading it, also when kernel timing is
    on and the hand-over events are the timing events."""
    import torch
    syn = pkg.synthetic
    segs = [syn.make_segment(700 + s, 1500, 32, 12, 26) for s in range(64)]
    T = [s[0].shape[0] for s in segs]
    C = [len(s[1]) for s in segs]
    U = [len(s[2]) - 1 for s in segs]
    plan = engine.plan(pkg.CtcSegmentationParameters(index_duration=DUR).to_native(), 32, T, C, U)
    plan.set_timing(8)
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    d_lpz = torch.from_numpy(np.concatenate([s[0].reshape(-1) for s in segs])).to(dev)
    d_lab = torch.from_numpy(np.concatenate([s[1] for s in segs]).astype(np.int32)).to(dev)
    d_ub = torch.from_numpy(np.concatenate([s[2] for s in segs]).astype(np.int32)).to(dev)

    def run(pipelined):
        o = dict(fol=torch.zeros(sum(C), dtype=torch.int32, device=dev), cp=torch.zeros(sum(T), dtype=torch.float32, device=dev),
                 seg=torch.zeros(3, sum(U), dtype=torch.float64, device=dev), te=torch.zeros(64, dtype=torch.int32, device=dev),
                 status=torch.full((64,), -7, dtype=torch.int32, device=dev))
        plan.run_device(d_lpz.data_ptr(), d_lab.data_ptr(), d_ub.data_ptr(), o["fol"].data_ptr(), o["cp"].data_ptr(), None,
                        o["seg"][0].data_ptr(), o["seg"][1].data_ptr(), o["seg"][2].data_ptr(), o["te"].data_ptr(),
                        o["status"].data_ptr(), stream, pipelined=pipelined)
        return o

    ref = run(False)
    torch.cuda.synchronize()
    outs = [run(True), run(False), run(True), run(True), run(False)]
    plan.flush(stream)
    torch.cuda.synchronize()
    for o in outs:
        for k in ("status", "te", "fol", "seg"):
            assert torch.equal(o[k], ref[k]), k
    plan.close()


def test_one_over_long_text_is_that_segments_status(pkg, oracle, engine):
    """A segment with more label columns than one fill workgroup covers (> ~5 400) gets status 4 and
    a NotImplementedError of its own; the other segments of the batch are aligned (the reference would
    only be slow there; a batch-wide failure would lose every file of a lockstep round)."""
    syn = pkg.synthetic
    rng = np.random.default_rng(5)
    ok = [syn.make_segment(50 + s, 400, 32, 3, 20) for s in range(3)]
    gt, ub = syn.make_labels(rng, 200, 29, 32)          # 6 002 label columns
    long_seg = (syn.make_emissions(rng, 7000, 32, gt), gt, ub)
    short_audio = syn.make_segment(60, 30, 32, 4, 20)   # C > T: status 1, and no part in the launch shape
    segs = [ok[0], long_seg, ok[1], short_audio, ok[2]]
    res = _run(pkg, segs)
    assert [r["status"] for r in res] == [0, 4, 0, 1, 0]
    _check(pkg, oracle, [segs[i] for i in (0, 2, 4)], [res[i] for i in (0, 2, 4)])
    with pytest.raises(NotImplementedError):
        pkg.ctc_segmentation._raise_for_status(res[1]["status"])
    with pytest.raises(AssertionError):
        pkg.ctc_segmentation._raise_for_status(res[3]["status"])


@pytest.mark.parametrize("V", [32, 38, 64])
def test_blank_transition_cost_zero(pkg, oracle, V):
    """gratis_blank / config.blank_transition_cost_zero -- the knob both test scripts of the reference
    set (src/test/test_ctc_segmentation.py:31, src/test/test_seq2seq_segmentation.py:24): a column
    labelled blank (every utterance separator) stays for free in the fill, while the backtrack still
    infers transitions against max(blank, label).  Also with test_ctc_segmentation.py's other knobs
    (backtrack_from_max_t), in the windowed regime and with texts that share a fill."""
    syn = pkg.synthetic
    rng = np.random.default_rng(77 + V)
    segs = []
    for s in range(10):
        T = int(rng.integers(50, 1300))
        U = int(rng.integers(1, 8))
        n = int(rng.integers(2, max(3, min(30, (T - 3) // (U + 1) - 1))))
        segs.append(syn.make_segment(1200 + s + V, T, V, U, n))
    kw = dict(blank_transition_cost_zero=True)
    _check(pkg, oracle, segs, _run(pkg, segs, **kw), cfg_kw=kw)
    kw2 = dict(blank_transition_cost_zero=True, backtrack_from_max_t=True)
    _check(pkg, oracle, segs, _run(pkg, segs, **kw2), cfg_kw=kw2)
    kw3 = dict(blank_transition_cost_zero=True, preamble_transition_cost_zero=False)
    _check(pkg, oracle, segs, _run(pkg, segs, **kw3), cfg_kw=kw3)
    kw4 = dict(blank_transition_cost_zero=True, min_window_size=120, max_window_size=4000)
    _check(pkg, oracle, segs, _run(pkg, segs, **kw4), cfg_kw=kw4)
    # the window's text and the text minus its last utterances over the same emissions: one fill
    lpz, gt, ub = syn.make_segment(1300 + V, 700, V, 6, 18)
    members = [(lpz, gt[:ub[k] + 1].copy(), ub[:k + 1].copy()) for k in range(6, 0, -1)]
    _check(pkg, oracle, members, _run(pkg, members, **kw), cfg_kw=kw)


@pytest.mark.parametrize("V", [100, 400])
def test_blank_transition_cost_zero_over_more_than_63_distinct_labels_is_refused(pkg, V):
    """(checkpoint mode -- the only one that takes the flag -- stages at most 64 vocabulary columns: a launch that looks
    at more distinct labels than that is refused, whatever the size of the vocabulary)"""
    rng = np.random.default_rng(3)
    gt = np.r_[-1, np.arange(1, 80)].astype(np.int64)
    lpz = pkg.synthetic.make_emissions(rng, 300, V, gt, blank=0)
    with pytest.raises(NotImplementedError):
        _run(pkg, [(lpz, gt, np.array([1, len(gt) - 1]))], blank_transition_cost_zero=True)


def _bursty_segment(seed, T, C, V=32, blank=0):
    """Emissions with a planted path that alternates runs of a SWITCH in every frame with long stays (and a long
    wait in the start column): the slope of the path changes abruptly from block to block -- what the speculative
    windows of the checkpoint-mode backtrack (stride_backtrack_kernel) predict worst."""
    rng = np.random.default_rng(seed)
    gt = np.concatenate([[-1], rng.integers(1, V, size=C - 2), [blank]]).astype(np.int64)
    firsts, t, c = [], int(rng.integers(1, max(2, T // 6))), 1
    while c < C:
        run = int(rng.integers(8, 70))                   # columns entered in consecutive frames
        for _ in range(min(run, C - c)):
            firsts.append(t)
            t += 1
            c += 1
        t += int(rng.integers(20, 160))                  # ... then a long stay
    firsts = np.asarray(firsts)
    firsts = np.minimum(firsts, T - 1 - (C - 1 - np.arange(1, C))[::-1] * 0)   # keep them inside the audio
    if firsts[-1] >= T:
        firsts = (firsts.astype(np.float64) * (T - 2) / firsts[-1]).astype(np.int64) + 1
        firsts = np.maximum.accumulate(np.maximum(firsts, np.arange(1, C)))
        firsts = np.minimum(firsts, T - 1 - (C - 1 - np.arange(1, C)))
    col = np.zeros(T, np.int64)
    col[firsts] += 1
    col = np.cumsum(col)
    logits = (2.0 * rng.standard_normal((T, V))).astype(np.float32)
    sym = np.where(col > 0, gt[np.minimum(np.maximum(col, 1), C - 1)], blank)
    is_first = np.zeros(T, bool)
    is_first[firsts] = True
    logits[np.arange(T), np.where(is_first, sym, np.where(rng.random(T) < 0.3, sym, blank))] += np.float32(9.0)
    z = logits - logits.max(axis=1, keepdims=True)
    lpz = (z - np.log(np.exp(z.astype(np.float64)).sum(axis=1, keepdims=True)).astype(np.float32)).astype(np.float32)
    ub = np.array([1, C - 1], np.int64)
    return lpz, gt, ub


@pytest.mark.parametrize("striders,windows", [(3, 2), (5, 1), (7, 3), (1, 1)])
def test_checkpoint_backtrack_survives_paths_its_speculation_cannot_predict(pkg, oracle, monkeypatch, striders, windows):
    """Bursty paths (runs of one SWITCH per frame, then long stays), more striders than shipped (deeper
    speculation), one / three windows per block: every miss is recomputed from the exact entry column, the results
    are the oracle's."""
    monkeypatch.setenv("CTCFA_CHECKPOINT", "1")
    monkeypatch.setenv("CTCFA_SB_WAVES", str(striders))
    monkeypatch.setenv("CTCFA_SB_WINDOWS", str(windows))
    segs = [_bursty_segment(100 + s, T, C) for s, (T, C) in
            enumerate([(1500, 420), (2400, 700), (900, 300), (3000, 640), (700, 96), (2000, 1100)])]
    _check(pkg, oracle, segs, _run(pkg, segs))
