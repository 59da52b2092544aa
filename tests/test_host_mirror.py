"""CPU suite: the host mirror of the aligner protocol (no DP call, no GPU)."""
import numpy as np
import pytest

from oracle import ctc_segmentation_twin as tw
from tests.fakes import FakeASR


def test_parameters_defaults_and_flags(pkg):
    c = pkg.CtcSegmentationParameters()
    ref = tw.CtcSegmentationParameters()
    for k in ("max_prob", "skip_prob", "min_window_size", "max_window_size", "index_duration",
              "score_min_mean_over_L", "space", "blank", "replace_spaces_with_blanks",
              "blank_transition_cost_zero", "preamble_transition_cost_zero", "backtrack_from_max_t",
              "self_transition", "start_of_ground_truth", "excluded_characters", "tokenized_meta_symbol"):
        assert getattr(c, k) == getattr(ref, k), k
    assert c.flags == 2
    c.set(blank_transition_cost_zero=True)
    assert c.flags == 3
    with pytest.raises(ValueError):
        c.set(not_a_field=1)
    c2 = pkg.CtcSegmentationParameters(subsampling_factor=4, frame_duration_ms=10)
    assert c2.index_duration_in_seconds == 0.04


def test_prepare_token_list_matches_twin(pkg):
    rng = np.random.default_rng(0)
    for _ in range(10):
        utts = [rng.integers(0, 30, size=int(rng.integers(0, 8))) for _ in range(int(rng.integers(1, 5)))]
        a = pkg.prepare_token_list(pkg.CtcSegmentationParameters(), utts)
        b = tw.prepare_token_list(tw.CtcSegmentationParameters(), utts)
        assert np.array_equal(a[0], b[0]) and a[1] == b[1]
        assert a[0].dtype == np.int64 and a[0].shape[1] == 1


def test_prepare_text_matches_twin(pkg):
    chars = ["<blank>", "<unk>", "▁", "a", "b", "c", "ab", "▁a", "·"]
    text = ["ab ca", "b", "a.b,c"]
    for rs in (False, True):
        a = pkg.prepare_text(pkg.CtcSegmentationParameters(char_list=list(chars), replace_spaces_with_blanks=rs), text)
        b = tw.prepare_text(tw.CtcSegmentationParameters(char_list=list(chars), replace_spaces_with_blanks=rs), text)
        assert np.array_equal(a[0], b[0]) and a[1] == b[1]
    assert a[0].shape[1] == 7  # max piece length ("<blank>") sets the span dimension S


def test_task_str_format(pkg):
    t = pkg.CTCSegmentationTask(name="utt_7", text=["HOLA QUE TAL", "ADIOS"],
                                segments=[(0.0, 1.234, -0.12345), (1.239, 2.5, -10000000000.0)])
    lines = str(t).split("\n")
    assert lines[0] == "utt_7_0000 utt_7 0.00 1.23 -0.1235 HOLA QUE TAL"   # fields [2],[3],[4],[-1] after split(" ", 5)
    assert lines[1] == "utt_7_0001 utt_7 1.24 2.50 -10000000000.0000 ADIOS"
    assert lines[2] == ""
    t2 = pkg.CTCSegmentationTask(name="n", text=["x"], segments=[(0, 1, -1)], utt_ids=["id0"])
    assert str(t2) == "id0 n 0.00 1.00 -1.0000 x\n"


def test_aligner_protocol_until_the_dp(pkg):
    asr = FakeASR(seed=1)
    al = pkg.CTCSegmentation(asr, kaldi_style_text=False, time_stamps="fixed", scoring_length=30)
    assert al.config.score_min_mean_over_L == 30 and al.kaldi_style_text is False
    assert al.config.char_list[0] == "<blank>" and len(al.config.char_list) == 32
    ratio = al.estimate_samples_to_frames_ratio()
    assert ratio == 215040 / 671                       # wav2vec2 geometry of the fake encoder
    import torch
    audio = torch.zeros(16000 * 3)
    lpz = al.get_lpz(audio)
    assert lpz.shape == ((48000 - 400) // 320 + 1, 32) and lpz.dtype == np.float32
    np.testing.assert_allclose(np.exp(lpz).sum(1), 1.0, rtol=1e-5)
    # list input (iterative / word level) and single-string input (search_on_speech.py:77-82)
    task = al.prepare_segmentation_task(["HOLA", "", "QUE TAL"], lpz, "utt", 48000)
    assert task.text == ["HOLA", "QUE TAL"] and task.utt_ids is None
    assert task.config.index_duration == ratio / 16000
    gt = task.ground_truth_mat[:, 0].tolist()
    assert gt[0] == -1 and gt[1] == 0 and gt[-1] == 0 and task.utt_begin_indices[0] == 1
    assert len(task.utt_begin_indices) == 3
    task2 = al.prepare_segmentation_task("·HOLA·", lpz, "utt", 48000)
    assert task2.text == ["·HOLA·"]
    # <unk> ids (the "·") are filtered out of the label sequence
    assert 1 not in task2.ground_truth_mat[:, 0].tolist()
    # kaldi style
    al2 = pkg.CTCSegmentation(asr, kaldi_style_text=True)
    ids, text = al2._split_text("u1 HOLA\nu2 QUE TAL\nbroken")
    assert ids == ["u1", "u2"] and text == ["HOLA", "QUE TAL"]
    with pytest.raises(NotImplementedError):
        pkg.CTCSegmentation(asr, time_stamps="nope")
    kw = dict(min_window_size=80000, max_window_size=100000, gratis_blank=False, set_blank=0)  # test_ctc_segmentation.py:20-25
    al3 = pkg.CTCSegmentation(asr, kaldi_style_text=False, **kw)
    assert al3.config.min_window_size == 80000 and al3.config.blank == 0


def test_batched_emissions_one_padded_forward(pkg):
    """get_lpz_batch: one padded encoder call; with a convolution-only encoder every item's
    frames equal the batch-of-1 result the reference computes."""
    import torch
    asr = FakeASR(seed=2)
    calls = []
    enc = asr.encode_batch
    asr.encode_batch = lambda wavs, lens: (calls.append((tuple(wavs.shape), lens.tolist())), enc(wavs, lens))[1]
    al = pkg.CTCSegmentation(asr, kaldi_style_text=False, time_stamps="fixed")
    g = torch.Generator().manual_seed(5)
    waves = [torch.randn(n, generator=g) for n in (48000, 16000 * 7 + 123, 400, 31999)]
    singles = [al.get_lpz(w) for w in waves]
    calls.clear()
    batch = al.get_lpz_batch(waves, frames_fn=pkg.wav2vec2_frames)
    assert len(calls) == 1 and calls[0][0] == (4, 16000 * 7 + 123)
    np.testing.assert_allclose(calls[0][1], [n / (16000 * 7 + 123) for n in (48000, 16000 * 7 + 123, 400, 31999)], rtol=1e-6)
    for s, b in zip(singles, batch):
        assert s.shape == b.shape and b.dtype == np.float32
        np.testing.assert_allclose(b, s, rtol=2e-6, atol=2e-6)  # the batched matmul blocks differently: a few ulp
    # default frame rule (SpeechBrain's relative lengths) is within one frame of the exact count
    for s, b in zip(singles, al.get_lpz_batch(waves)):
        assert abs(b.shape[0] - s.shape[0]) <= 1


def test_inputs_numpy_would_refuse_are_refused_on_the_host(pkg):
    """Labels and utterance starts reach the kernels as raw indices: what the package's NumPy code
    would answer with IndexError is refused before any launch (no GPU needed for the refusal)."""
    cs = pkg.ctc_segmentation
    lpz = np.zeros((20, 8), np.float32)
    good = np.array([-1, 0, 3, 4, 0], np.int32)
    cs._validate_segments([lpz], [good], [np.array([1, 4])])
    with pytest.raises(IndexError):
        cs._validate_segments([lpz], [np.array([-1, 0, 8, 0], np.int32)], None)          # label id == V
    with pytest.raises(IndexError):
        cs._validate_segments([lpz], [np.array([-1, 0, -3, 0], np.int32)], None)         # negative label
    with pytest.raises(ValueError):
        cs._validate_segments([lpz, np.zeros((5, 9), np.float32)], [good, good], None)   # two vocabulary sizes
    with pytest.raises(ValueError):
        cs._validate_segments([lpz], [np.array([0, 1, 2], np.int32)], None)              # no leading -1
    with pytest.raises(IndexError):
        cs._validate_segments([lpz], [good], [np.array([1, 4, 4])])                      # trailing empty utterance
    with pytest.raises(IndexError):
        cs._validate_segments([lpz], [good], [np.array([0, 4])])                         # start column as an utterance start
    # per segment: one refused segment is its own error, not the launch's; a -1 inside a sequence is no error
    # (the package skips such a column: served by the label-matrix kernel)
    errors, minus_one = cs._segment_errors([lpz, lpz, lpz], [good, np.array([-1, 0, 8, 0], np.int32), np.array([-1, 0, -1, 3, 0], np.int32)],
                                           [np.array([1, 4]), np.array([1, 3]), np.array([1, 4])])
    assert errors[0] is None and isinstance(errors[1], IndexError) and errors[2] is None
    assert minus_one == [False, False, True]
    # a malformed or odd FIRST segment is that segment's error too: the launch's width is the one most segments agree on
    errors, _ = cs._segment_errors([np.zeros(7, np.float32), lpz, lpz], [good, good, good], None)
    assert isinstance(errors[0], ValueError) and errors[1] is None and errors[2] is None
    errors, _ = cs._segment_errors([np.zeros((5, 9), np.float32), lpz, lpz], [good, good, good], None)
    assert isinstance(errors[0], ValueError) and errors[1] is None and errors[2] is None


def test_every_task_keeps_its_own_timing_config(pkg):
    """time_stamps="auto": index_duration = speech_len / lpz_len / fs differs from task to task; tasks
    prepared up front (run_batched, align_words) must not all see the last one's value."""
    from tests.fakes import FakeASR
    al = pkg.CTCSegmentation(FakeASR(seed=1), kaldi_style_text=False, time_stamps="auto")
    a = al.prepare_segmentation_task(["HOLA"], np.zeros((50, 32), np.float32), "a", 16000)
    b = al.prepare_segmentation_task(["HOLA"], np.zeros((50, 32), np.float32), "b", 32000)
    assert a.config is not b.config
    assert a.config.index_duration == pytest.approx(16000 / 50 / 16000)
    assert b.config.index_duration == pytest.approx(32000 / 50 / 16000)


def test_synthetic_labels_over_an_alphabet(pkg):
    """bench.py --alphabet: texts that use the first N non-blank entries of a larger vocabulary."""
    import numpy as np
    rng = np.random.default_rng(3)
    gt, ub = pkg.synthetic.make_labels(rng, 4, 50, 38, blank=0, alphabet=28)
    assert gt[0] == -1 and gt[1:].min() >= 0 and gt[1:].max() <= 28
    assert len(np.unique(gt[1:])) > 20 and len(ub) == 5
    lpz, gt2, ub2 = pkg.synthetic.make_segment(7, 400, 38, 3, 20, alphabet=28)
    assert lpz.shape == (400, 38) and gt2[1:].max() <= 28
