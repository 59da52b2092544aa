"""GPU parity for LABEL MATRICES (multi-character tokens, S > 1): ground_truth_mat as the "classic"
text converter of SpeechBrain's CTCSegmentation builds it (ctc_segmentation.prepare_text) through
ctcfa_align_batch_spans, against the oracle's cython_fill_table / backtrack restatement with S > 1."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DUR = 320.4769 / 16000
SCORE_TOL = 1e-4

# a sub-word vocabulary over a small alphabet: single characters and tokens of 2-5 characters
CHARS = ["<blank>", "<unk>"] + list("ABCDEFGHIJKLMNOP") + ["AB", "CD", "ABC", "EF", "GHI", "JKLM", "NOP", "BA", "DC", "ABCDE", "PA", "FE"]


def _texts(rng, n_utts, n_chars):
    """Utterances put together from tokens of the list above, so that multi-character tokens do end in
    many columns (and overlap: "ABCDE" also reads A-B-C-D-E, AB-CD-E, ABC-DC...)."""
    pieces = CHARS[2:]
    out = []
    for _ in range(n_utts):
        u = ""
        want = int(rng.integers(2, n_chars + 1))
        while len(u) < want:
            u += pieces[int(rng.integers(0, len(pieces)))]
        out.append(u)
    return out


def _segment(pkg, seed, T, n_utts, n_chars, sharp=3.0):
    """(lpz, ground_truth_mat [C, S], utt_begin): emissions that loosely follow a random token path."""
    rng = np.random.default_rng(seed)
    cfg = pkg.CtcSegmentationParameters(char_list=list(CHARS), index_duration=DUR)
    mat, ub = pkg.prepare_text(cfg, _texts(rng, n_utts, n_chars))
    V = len(CHARS)
    logits = rng.normal(0.0, 1.0, size=(T, V)).astype(np.float32)
    # favour, frame by frame, some token that ends at the column a uniform walk would be in
    C = len(mat)
    for t in range(T):
        c = min(C - 1, int(t * C / T))
        cand = [int(j) for j in mat[c] if j >= 0]
        logits[t, cand[int(rng.integers(0, len(cand)))] if cand else 0] += sharp
    lpz = (logits - np.log(np.exp(logits).sum(axis=1, keepdims=True))).astype(np.float32)
    return lpz, mat, np.asarray(ub, np.int64)


def _check(oracle, segs, res, **cfg_kw):
    ocfg = oracle.make_config(index_duration=DUR, **cfg_kw)
    for i, ((lpz, mat, ub), r) in enumerate(zip(segs, res)):
        o = oracle.get_segments(lpz, mat, ub, ocfg)
        assert r["status"] == o["status"], (i, r["status"], o["status"])
        if o["status"] != 0:
            continue
        assert r["t_end"] == o["t_end"], i
        assert np.array_equal(r["frame_of_label"], o["frame_of_label"]), f"segment {i}: frame indices differ"
        assert np.array_equal(r["char_prob"].astype(np.float64), o["char_probs"]), f"segment {i}: char_probs"
        assert np.array_equal(r["state"], o["state"]), f"segment {i}: state list"
        assert np.array_equal(r["seg_start"], o["seg_start"]) and np.array_equal(r["seg_end"], o["seg_end"]), i
        np.testing.assert_allclose(r["seg_score"], o["seg_score"], rtol=0, atol=SCORE_TOL)


def _run(pkg, segs, **cfg):
    config = pkg.CtcSegmentationParameters(index_duration=DUR, **cfg)
    return pkg.ctc_segmentation.get_segments_device(config, [s[0] for s in segs], [s[1] for s in segs],
                                                    [s[2] for s in segs])


def test_label_matrices_match_the_oracle(pkg, oracle):
    rng = np.random.default_rng(3)
    segs = [_segment(pkg, 100 + i, int(rng.integers(40, 700)), int(rng.integers(1, 6)), int(rng.integers(3, 14)))
            for i in range(12)]
    assert max(int((s[1][:, 1:] != -1).sum()) for s in segs) > 0     # multi-character tokens do occur
    res = _run(pkg, segs)
    _check(oracle, segs, res)
    multi = 0
    for (lpz, mat, ub), r in zip(segs, res):   # and some path really takes a multi-character token
        st = r["state"]
        multi += int(np.isin(st[st >= 0], np.arange(18, len(CHARS))).sum())
    assert multi > 0


def test_label_matrices_other_knobs_and_windows(pkg, oracle):
    segs = [_segment(pkg, 200 + i, T, U, n) for i, (T, U, n) in enumerate([(300, 3, 9), (520, 5, 12), (90, 1, 6), (800, 4, 13)])]
    for kw in (dict(backtrack_from_max_t=True), dict(preamble_transition_cost_zero=False),
               dict(blank_transition_cost_zero=True), dict(min_window_size=100, max_window_size=3000), dict(blank=0)):
        _check(oracle, segs, _run(pkg, segs, **kw), **kw)


def test_text_longer_than_audio_with_label_matrices(pkg, oracle):
    lpz, mat, ub = _segment(pkg, 300, 200, 4, 12)
    segs = [(lpz[:20], mat, ub), _segment(pkg, 301, 150, 2, 8)]
    res = _run(pkg, segs)
    assert res[0]["status"] == 1 and res[1]["status"] == 0
    _check(oracle, segs, res)


def test_classic_text_converter_end_to_end(pkg, oracle):
    """CTCSegmentation(text_converter="classic") -- prepare_text's label matrix -- through get_segments:
    same segments as the oracle-backed aligner."""
    from tests.fakes import FakeASR, oracle_backed

    asr = FakeASR(seed=11)
    lpz = None
    for mk in (lambda: pkg.CTCSegmentation(asr, kaldi_style_text=False, text_converter="classic", time_stamps="fixed"),):
        hip, ref = mk(), oracle_backed(mk(), oracle)
        import torch
        speech = torch.randn(16000 * 6, generator=torch.Generator().manual_seed(4)) * 0.1
        lpz = hip.get_lpz(speech)
        text = ["HOLA QUE TAL", "BUENOS DIAS"]
        ta = hip.prepare_segmentation_task(text, lpz, "x", speech.shape[0])
        tb = ref.prepare_segmentation_task(text, lpz, "x", speech.shape[0])
        assert np.asarray(ta.ground_truth_mat).ndim == 2
        a, b = hip.get_segments(ta), ref.get_segments(tb)
        assert [s[:2] for s in a["segments"]] == [s[:2] for s in b["segments"]]
        np.testing.assert_allclose([s[2] for s in a["segments"]], [s[2] for s in b["segments"]], rtol=0, atol=SCORE_TOL)


def test_label_width_limits(pkg, oracle):
    """Sixteen characters per token is what the kernel takes; wider matrices are refused before the launch."""
    rng = np.random.default_rng(8)
    V = 40
    T, C, S = 300, 40, 16
    mat = np.full((C, S), -1, np.int64)
    mat[1:, 0] = rng.integers(1, V, size=C - 1)
    mat[0] = -1
    for c in range(S, C, 5):           # a few long tokens, up to 16 characters
        mat[c, int(rng.integers(1, S))] = int(rng.integers(1, V))
    mat[C - 1, :] = -1
    mat[C - 1, 0] = 0                  # closing blank
    ub = np.asarray([1, C - 1], np.int64)
    mat[1, 0] = 0
    logits = rng.normal(size=(T, V)).astype(np.float32)
    lpz = (logits - np.log(np.exp(logits).sum(axis=1, keepdims=True))).astype(np.float32)
    segs = [(lpz, mat, ub)]
    _check(oracle, segs, _run(pkg, segs))
    wide = np.concatenate([mat, np.full((C, 1), -1, np.int64)], axis=1)
    wide[20, 16] = 3
    with pytest.raises(NotImplementedError):
        _run(pkg, [(lpz, wide, ub)])
