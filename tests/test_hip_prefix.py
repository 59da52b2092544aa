"""GPU parity of SHARED FILLS (ctcfa_align_batch_shared / ctcfa_plan_create_shared): segments over the
same emissions whose texts are prefixes of one another -- the repeat loop of the anchor iteration
(iterative_utterance_alignment.py:201-219, drops at :283,352,374) -- are served by one trellis fill.
Every member must equal an independent oracle run on its own (emissions, text[:k])."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SCORE_TOL = 1e-4
DUR = 320.4769 / 16000
FUZZ_SALT = int(__import__("os").environ.get("CTCFA_FUZZ_SALT", "0"))  # soak runs: other random cases


@pytest.fixture(autouse=True, params=["auto", "checkpoint", "decision"])
def dp_mode(request, monkeypatch):
    if request.param == "checkpoint":
        monkeypatch.setenv("CTCFA_CHECKPOINT", "1")
    if request.param == "decision":
        monkeypatch.setenv("CTCFA_DECISION_BITS", "1")
    return request.param


def prefixes(seg, keep=None):
    """(lpz, gt, ub) -> the segment and its texts minus the last 1, 2, ... utterances (same lpz object)."""
    lpz, gt, ub = seg
    U = len(ub) - 1
    ks = list(range(U, 0, -1)) if keep is None else [k for k in keep if 1 <= k <= U]
    return [(lpz, gt[:ub[k] + 1].copy(), ub[:k + 1].copy()) for k in ks]


def check(oracle, segs, res, **cfg_kw):
    ocfg = oracle.make_config(index_duration=DUR, **cfg_kw)
    for i, ((lpz, gt, ub), r) in enumerate(zip(segs, res)):
        o = oracle.get_segments(lpz, gt, ub, ocfg)
        assert r["status"] == o["status"], (i, r["status"], o["status"])
        if o["status"] != 0:
            continue
        assert r["t_end"] == o["t_end"], (i, len(gt))
        assert np.array_equal(r["frame_of_label"], o["frame_of_label"]), f"member {i}: frame indices differ"
        assert np.array_equal(r["char_prob"].astype(np.float64), o["char_probs"]), f"member {i}: char_probs"
        assert np.array_equal(r["state"], o["state"]), f"member {i}: state list"
        assert np.array_equal(r["seg_start"], o["seg_start"]) and np.array_equal(r["seg_end"], o["seg_end"]), i
        np.testing.assert_allclose(r["seg_score"], o["seg_score"], rtol=0, atol=SCORE_TOL)


def run(pkg, segs, **cfg):
    config = pkg.CtcSegmentationParameters(index_duration=DUR, **cfg)
    return pkg.ctc_segmentation.get_segments_device(config, [s[0] for s in segs], [s[1] for s in segs],
                                                    [s[2] for s in segs])


@pytest.mark.parametrize("V", [32, 38, 64, 100])
def test_every_prefix_equals_its_own_oracle_run(pkg, oracle, V):
    syn = pkg.synthetic
    rng = np.random.default_rng(V + 1000 * FUZZ_SALT)
    segs = []
    for g in range(10):
        T = int(rng.integers(60, 1200))
        U = int(rng.integers(1, 9))
        n = int(rng.integers(2, max(3, min(30, (T - 3) // (U + 1) - 1))))
        members = prefixes(syn.make_segment(500 + g + V + 77 * FUZZ_SALT, T, V, U, n))
        if g % 3 == 0:   # the longest member need not come first
            members = members[::-1]
        segs += members
        if g % 4 == 1:   # segments of their own in between
            segs.append(syn.make_segment(900 + g, int(rng.integers(40, 500)), V, 2, 9))
    assert pkg.ctc_segmentation.shared_emissions([s[0] for s in segs]) is not None
    check(oracle, segs, run(pkg, segs))


@pytest.mark.parametrize("T", [33, 65, 257, 258, 2, 34])
def test_last_row_of_a_watch_column(pkg, oracle, T):
    """(T - 1) % 32 == 0: the final table row lives outside the last full 32-row block."""
    syn = pkg.synthetic
    segs = []
    for s in range(3):
        U = 1 if T < 20 else 3
        n = 1 if T < 20 else max(1, min(12, T // 6))
        segs += prefixes(syn.make_segment(40 + s + T, T, 32, U, n))
    check(oracle, segs, run(pkg, segs))


def test_from_max_t_and_plain_preamble(pkg, oracle):
    syn = pkg.synthetic
    segs = []
    for s in range(4):
        segs += prefixes(syn.make_segment(70 + s, 300 + 40 * s, 32, 4, 15))
    check(oracle, segs, run(pkg, segs, backtrack_from_max_t=True), backtrack_from_max_t=True)
    check(oracle, segs, run(pkg, segs, preamble_transition_cost_zero=False), preamble_transition_cost_zero=False)


def test_leader_with_text_longer_than_audio(pkg, oracle):
    """The full text does not fit the audio (status 1) but its prefixes do: they still share a fill,
    led by the longest member that is aligned."""
    syn = pkg.synthetic
    lpz, gt, ub = syn.make_segment(11, 400, 32, 6, 30)
    short = lpz[:150]
    segs = prefixes((short, gt, ub))
    res = run(pkg, segs)
    assert [r["status"] for r in res][:2] == [1, 1] and res[-1]["status"] == 0
    check(oracle, segs, res)


@pytest.mark.parametrize("V,alphabet", [(38, 28), (200, 28), (100, 50)])
def test_shared_fills_of_a_narrowed_plan(pkg, oracle, engine, V, alphabet):
    """A narrowed plan (the 32- or the 64-entry ring) shares fills like any other: the leader's ring holds every label of
    its prefixes; each member's backtrack derives the ring of its own text."""
    syn = pkg.synthetic
    segs = []
    for g in range(6):
        segs += prefixes(syn.make_segment(3100 + g + V, 700 + 90 * g, V, 6 + g % 3, 18, alphabet=alphabet))
    check(oracle, segs, run(pkg, segs))
    config = pkg.CtcSegmentationParameters(index_duration=DUR)
    first = {}
    emission_of = [first.setdefault(id(s[0]), b) for b, s in enumerate(segs)]
    plan = engine.plan(config.to_native(), V, [s[0].shape[0] for s in segs], [len(s[1]) for s in segs], [len(s[2]) - 1 for s in segs],
                       emission_of=emission_of, labels=np.concatenate([s[1] for s in segs]))
    fills, blocks = plan.sharing()
    assert blocks == 6 and fills == 6
    assert plan.info["vocab_pitch"] == (34 if alphabet <= 31 else 66)
    plan.close()


def test_members_that_are_not_prefixes_are_filled_by_themselves(pkg, oracle, engine):
    syn = pkg.synthetic
    lpz, gt, ub = syn.make_segment(21, 500, 32, 5, 20)
    _, gt2, ub2 = syn.make_segment(22, 500, 32, 3, 25)       # another text over the same emissions
    segs = prefixes((lpz, gt, ub)) + [(lpz, gt2, ub2), (lpz, gt.copy(), ub.copy())] + prefixes((lpz, gt2, ub2))[1:]
    check(oracle, segs, run(pkg, segs))
    config = pkg.CtcSegmentationParameters(index_duration=DUR)
    plan = engine.plan(config.to_native(), 32, [500] * len(segs), [len(s[1]) for s in segs], [len(s[2]) - 1 for s in segs],
                       emission_of=[0] * len(segs), labels=np.concatenate([s[1] for s in segs]))
    fills, blocks = plan.sharing()
    # group of gt: 5 members, one fill; the duplicate of the full text: its own; gt2 (3 utterances): the
    # longest text decides the leader, so gt2 and its two prefixes are not prefixes of it: three more
    assert blocks == 1 and fills == 1 + 1 + 3
    plan.close()


def test_more_prefixes_than_watch_columns(pkg, oracle):
    syn = pkg.synthetic
    segs = prefixes(syn.make_segment(31, 900, 32, 22, 6))
    assert len(segs) == 22
    check(oracle, segs, run(pkg, segs))


def test_one_token_utterances(pkg, oracle):
    """Watch columns two label columns apart."""
    syn = pkg.synthetic
    rng = np.random.default_rng(5)
    gt, ub = syn.make_labels_from_lengths(rng, [1, 1, 2, 1, 3, 1, 1], 32)
    lpz = syn.make_emissions(np.random.default_rng(6), 120, 32, gt, 0)
    segs = prefixes((lpz, gt, ub))
    check(oracle, segs, run(pkg, segs))


def _plan_run(pkg, engine, segs, emission_of, K, vocab=32):
    import torch

    config = pkg.CtcSegmentationParameters(index_duration=DUR)
    T = [s[0].shape[0] for s in segs]
    C = [len(s[1]) for s in segs]
    U = [len(s[2]) - 1 for s in segs]
    labels = np.concatenate([s[1] for s in segs]).astype(np.int32)
    plan = engine.plan(config.to_native(), vocab, T, C, U, force_cols_per_lane=K, emission_of=emission_of, labels=labels)
    dev = torch.device("cuda:0")
    own = [s[0].reshape(-1) for b, s in enumerate(segs) if emission_of[b] == b]
    d_lpz = torch.from_numpy(np.concatenate(own)).to(dev)
    d_lab = torch.from_numpy(labels).to(dev)
    d_ub = torch.from_numpy(np.concatenate([s[2] for s in segs]).astype(np.int32)).to(dev)
    nT, nC, nU = sum(T), sum(C), sum(U)
    d_fol = torch.empty(nC, dtype=torch.int32, device=dev)
    d_cp = torch.empty(nT, dtype=torch.float32, device=dev)
    d_st = torch.empty(nT, dtype=torch.int32, device=dev)
    d_seg = torch.empty(3, nU, dtype=torch.float64, device=dev)
    d_te = torch.empty(len(segs), dtype=torch.int32, device=dev)
    d_status = torch.empty(len(segs), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    for pipelined in (False, True, True, True):   # both entries; the pipelined one alternates workspaces
        plan.run_device(d_lpz.data_ptr(), d_lab.data_ptr(), d_ub.data_ptr(), d_fol.data_ptr(), d_cp.data_ptr(),
                        d_st.data_ptr(), d_seg[0].data_ptr(), d_seg[1].data_ptr(), d_seg[2].data_ptr(),
                        d_te.data_ptr(), d_status.data_ptr(), stream, pipelined=pipelined)
        plan.flush(stream)
        torch.cuda.synchronize()
    fol, cp, st = d_fol.cpu().numpy(), d_cp.cpu().numpy(), d_st.cpu().numpy()
    seg, te, status = d_seg.cpu().numpy(), d_te.cpu().numpy(), d_status.cpu().numpy()
    to = np.concatenate([[0], np.cumsum(T)])
    co = np.concatenate([[0], np.cumsum(C)])
    uo = np.concatenate([[0], np.cumsum(U)])
    res = [dict(status=int(status[b]), t_end=int(te[b]), frame_of_label=fol[co[b]:co[b + 1]],
                char_prob=cp[to[b]:to[b + 1]], state=st[to[b]:to[b + 1]], seg_start=seg[0][uo[b]:uo[b + 1]],
                seg_end=seg[1][uo[b]:uo[b + 1]], seg_score=seg[2][uo[b]:uo[b + 1]]) for b in range(len(segs))]
    sharing = plan.sharing()
    info = dict(plan.info)
    plan.close()
    return res, sharing, info


@pytest.mark.parametrize("K", [1, 2, 0])
def test_watch_columns_across_many_tiles(pkg, oracle, engine, K):
    """Texts of ~500 label columns: the watch columns lie in different tiles (and, for K = 2, at either
    k of their lane); dead-zone skipping has to keep every member's cells alive.  K = 0: the plan's choice
    (it may only pick tile widths that can watch)."""
    syn = pkg.synthetic
    segs, emission_of = [], []
    for g, (T, U, n) in enumerate([(1500, 12, 40), (700, 9, 33), (2100, 14, 37), (520, 10, 45)]):
        members = prefixes(syn.make_segment(600 + g, T, 32, U, n))
        first = len(segs)
        segs += members
        emission_of += [first] * len(members)
    res, (fills, blocks), info = _plan_run(pkg, engine, segs, emission_of, K)
    assert blocks == 4 and fills == 4
    assert info["cols_per_lane"] in (1, 2)
    check(oracle, segs, res)


def test_device_resident_emissions(pkg, oracle):
    import torch

    syn = pkg.synthetic
    segs, dev_segs = [], []
    for g in range(5):
        members = prefixes(syn.make_segment(700 + g, 200 + 90 * g, 32, 5, 14))
        d = torch.from_numpy(members[0][0]).to("cuda:0")
        segs += members
        dev_segs += [(d, m[1], m[2]) for m in members]
    check(oracle, segs, run(pkg, dev_segs))


def test_speculating_anchor_iteration_equals_the_plain_one(pkg, oracle):
    """The anchor iteration over several files and over the full sample file (BASELINE configs[1]), HIP
    aligner: speculate=n gives the rows of the plain driver with fewer launches, and what it computes
    ahead shares the window's emissions and fill."""
    import json
    import os

    from tests.fakes import FakeASR
    from tests.replay_common import REPLAY_AUDIO_SECONDS, REPLAY_PARAMS, NoiseAudio, replay_vad
    from tests.test_anchor import GOLD as ANCHOR_GOLD

    anchor = pkg.anchor
    asr = FakeASR(seed=5, sharp=6.0)
    files = [("data/x/f%d.wav" % i, 14 + 3 * i, 70.0 + 8 * i, 40 + i) for i in range(4)]
    vad = lambda secs: [dict(Start=0.0, End=secs * 0.45, Segment_Length=secs * 0.45),
                        dict(Start=secs * 0.5, End=secs - 1.0, Segment_Length=secs * 0.5 - 1.0)]
    params = anchor.AnchorParams(threshold=-6.0, short_utterance_len=12, max_words_sequence=6)
    mk = lambda: pkg.CTCSegmentation(asr, kaldi_style_text=False, time_stamps="fixed", scoring_length=30)

    def run_files(speculate, stats):
        cos = [anchor.file_alignment(asr, NoiseAudio(secs, seed), path,
                                     [dict(r, Sample_Path=path) for r in ANCHOR_GOLD["tsv_rows"][:n]],
                                     vad(secs), 320.0, params) for path, n, secs, seed in files]
        return anchor.run_batched(cos, mk(), speculate=speculate, stats=stats)

    s0, s2 = {}, {}
    plain = run_files(0, s0)
    assert run_files(2, s2) == plain
    assert s2["answered_ahead"] > 0 and s2["rounds"] < s0["rounds"]

    rows = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "benedetti_rows.json")))["rows"]

    def run_replay(speculate, stats):
        co = anchor.file_alignment(FakeASR(seed=5, sharp=6.0), NoiseAudio(REPLAY_AUDIO_SECONDS, 2024), rows[0]["Sample_Path"],
                                   [dict(r) for r in rows], replay_vad(), 320.0, anchor.AnchorParams(**REPLAY_PARAMS))
        al = pkg.CTCSegmentation(FakeASR(seed=5, sharp=6.0), kaldi_style_text=False, time_stamps="fixed", scoring_length=30)
        return anchor.run_batched([co], al, speculate=speculate, stats=stats)[0]

    r0, r1 = {}, {}
    full = run_replay(0, r0)
    assert run_replay(1, r1) == full
    assert r1["requests"] == r0["requests"] and r1["launches"] + r1["answered_ahead"] == r0["launches"]
    assert r1["answered_ahead"] >= 10


def test_shared_fills_at_size_equal_separate_fills(pkg, oracle):
    """Groups at the sizes of BASELINE configs[2] (T = 3000, up to ~1500 label columns, up to 15 tiles):
    the shared call must give, member by member, exactly what separate emissions give (HIP against HIP, every
    member), and a sample of members is checked against the oracle."""
    syn = pkg.synthetic
    rng = np.random.default_rng(2024 + FUZZ_SALT)
    shared, separate = [], []
    for g in range(24):
        T = int(rng.integers(800, 3001))
        U = int(rng.integers(2, 40))
        n = int(rng.integers(8, max(9, min(40, (T - 3) // (U + 1) - 1))))
        if g == 0:   # one group of the full size whatever the draw
            T, U, n = 3000, 38, 38
        seg = syn.make_segment(4000 + g, T, 32, U, n)
        keep = sorted(set(int(k) for k in rng.integers(1, U + 1, size=int(rng.integers(1, 7)))) | {U}, reverse=True)
        members = prefixes(seg, keep=keep)
        shared += members
        separate += [(m[0].copy(), m[1], m[2]) for m in members]    # own emission arrays: nothing shared
    assert max(len(s[1]) for s in shared) > 1000
    a = run(pkg, shared)
    b = run(pkg, separate)
    for i, (x, y) in enumerate(zip(a, b)):
        assert x["status"] == y["status"] == 0, i
        assert x["t_end"] == y["t_end"], i
        for k in ("frame_of_label", "char_prob", "state", "seg_start", "seg_end", "seg_score"):
            assert np.array_equal(x[k], y[k]), (i, k)
    sample = list(range(0, len(shared), 7))
    check(oracle, [shared[i] for i in sample], [a[i] for i in sample])


def test_bad_sharing_arguments_are_refused(pkg, engine):
    syn = pkg.synthetic
    lpz, gt, ub = syn.make_segment(1, 200, 32, 3, 10)
    config = pkg.CtcSegmentationParameters(index_duration=DUR).to_native()
    lab = [gt.astype(np.int32), gt[:ub[2] + 1].astype(np.int32)]
    ubs = [ub, ub[:3]]
    with pytest.raises(ValueError):     # names a later segment
        engine.align_batch(config, [lpz, lpz], lab, ubs, emission_of=[1, 1])
    with pytest.raises(ValueError):     # the named segment has no emissions of its own
        engine.align_batch(config, [lpz, lpz, lpz], lab + [lab[1]], ubs + [ubs[1]], emission_of=[0, 0, 1])
    with pytest.raises(ValueError):     # another number of frames
        engine.align_batch(config, [lpz, lpz[:150]], lab, ubs, emission_of=[0, 0], shapes=[(200, 32), (150, 32)])
    with pytest.raises(ValueError):
        engine.align_batch(config, [lpz, lpz], lab, ubs, emission_of=[0])
    # the caller may vouch for the prefix property (labels=None): plan only
    plan = engine.plan(config, 32, [200, 200], [len(lab[0]), len(lab[1])], [3, 2], emission_of=[0, 0], labels=None)
    assert plan.sharing() == (1, 1)
    plan.close()
