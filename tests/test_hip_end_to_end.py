"""GPU: the caller loops (anchor iteration over several files in lockstep, word-level and
search rows) driven by the product aligner -- fake acoustic model on PyTorch, DP on the HIP
engine -- against the same loops driven by the oracle.  SURVEY.md §8 rows a1-a6 end to end."""
import importlib

import numpy as np
import pandas as pd
import pytest
import torch

from tests.fakes import FakeASR, oracle_backed
from tests.test_anchor import GOLD as ANCHOR_GOLD

pytestmark = pytest.mark.gpu


class NoiseAudio:
    """Deterministic pseudo-speech: seeded noise, so the fake encoder yields varied posteriors."""

    def __init__(self, seconds, seed, sr=16000):
        self.sample_rate, self.num_frames = sr, int(seconds * sr)
        g = torch.Generator().manual_seed(seed)
        self._data = torch.randn(self.num_frames, 1, generator=g) * 0.1

    def load(self, frame_offset, num_frames):
        return self._data[frame_offset:frame_offset + num_frames], self.sample_rate


def _rows(n):
    return [dict(r) for r in ANCHOR_GOLD["tsv_rows"][:n]]


def test_anchor_iteration_hip_equals_oracle(pkg, oracle):
    anchor = importlib.import_module(pkg.__name__ + ".anchor")
    asr = FakeASR(seed=5, sharp=6.0)
    files = [("data/x/f%d.wav" % i, 14 + 3 * i, 70.0 + 8 * i, 40 + i) for i in range(4)]
    vad = lambda secs: [dict(Start=0.0, End=secs * 0.45, Segment_Length=secs * 0.45),
                        dict(Start=secs * 0.5, End=secs - 1.0, Segment_Length=secs * 0.5 - 1.0)]
    params = anchor.AnchorParams(threshold=-6.0, short_utterance_len=12, max_words_sequence=6)

    def run(aligner):
        cos = [anchor.file_alignment(asr, NoiseAudio(secs, seed), path, [dict(r, Sample_Path=path) for r in _rows(n)],
                                     vad(secs), 320.0, params) for path, n, secs, seed in files]
        return anchor.run_batched(cos, aligner)

    hip = run(pkg.CTCSegmentation(asr, kaldi_style_text=False, time_stamps="fixed", scoring_length=30))
    ref = run(oracle_backed(pkg.CTCSegmentation(asr, kaldi_style_text=False, time_stamps="fixed", scoring_length=30), oracle))
    assert sum(len(r) for r in hip) > 20
    for h, r in zip(hip, ref):
        assert len(h) == len(r)
        for a, b in zip(h, r):
            assert a[:6] == b[:6] and a[7:] == b[7:]          # ids, times (10 ms strings -> floats), text
            assert abs(a[6] - b[6]) <= 1e-4 + 1e-9             # confidence score


def test_word_level_and_search_hip_equals_oracle(pkg, oracle):
    pl = importlib.import_module(pkg.__name__ + ".pipelines")
    tp = importlib.import_module(pkg.__name__ + ".text_prep")
    asr = FakeASR(seed=9, sharp=6.0)
    rows = []
    for i, r in enumerate(_rows(24)):
        norm = tp.normalize_transcript(str(r["Transcription"])).upper()
        words = norm.split(" ")
        rows.append(dict(r, Normalized_Transcription=norm, Wanted_Text=words[min(1, len(words) - 1)],
                         Start=float(i), End=float(i) + 4.0 + (i % 3)))
    df = pd.DataFrame(rows)
    audio = NoiseAudio(40.0, 77)
    opener = lambda path: audio
    mk = lambda: pkg.CTCSegmentation(asr, kaldi_style_text=False, time_stamps="fixed")
    hip = pl.align_words(asr, mk(), df, opener=opener)
    ref = pl.align_words(asr, oracle_backed(mk(), oracle), df, opener=opener)
    assert len(hip) == len(ref) >= 20
    for a, b in zip(hip, ref):
        assert a[:5] == b[:5] and a[6:] == b[6:] and abs(a[5] - b[5]) <= 1e-4 + 1e-9
    hip = pl.search_on_speech(asr, mk(), df, "horizonte", opener=opener)
    ref = pl.search_on_speech(asr, oracle_backed(mk(), oracle), df, "horizonte", opener=opener)
    assert len(hip) == len(ref) == len(df)
    for a, b in zip(hip, ref):
        assert a[:5] == b[:5] and a[6:] == b[6:] and abs(a[5] - b[5]) <= 1e-4 + 1e-9


def test_device_resident_emissions_match_the_oracle(pkg, oracle):
    """keep_lpz_on_device=True: encoder output stays in HBM (torch tensor) and feeds
    ctcfa_plan_run_device; results must equal the ORACLE's on the same matrices (and the
    host-array protocol path of the HIP engine)."""
    asr_gpu = FakeASR(seed=11, device="cuda:0")
    asr_cpu = FakeASR(seed=11)
    on_dev = pkg.CTCSegmentation(asr_gpu, kaldi_style_text=False, time_stamps="fixed", scoring_length=30,
                                 keep_lpz_on_device=True)
    on_host = pkg.CTCSegmentation(asr_cpu, kaldi_style_text=False, time_stamps="fixed", scoring_length=30)
    audio = NoiseAudio(30.0, 123)
    texts = [["HOLA QUE TAL", "MUY BIEN"], ["ADIOS"], ["UNO DOS TRES", "CUATRO", "CINCO SEIS"]]
    tasks_d, tasks_h = [], []
    for i, text in enumerate(texts):
        clip, sr = audio.load(16000 * 5 * i, 16000 * (6 + i))
        wav = asr_cpu.audio_normalizer(clip, sr)
        lpz_d = on_dev.get_lpz(wav)
        lpz_h = on_host.get_lpz(wav)
        assert lpz_d.is_cuda and isinstance(lpz_h, np.ndarray)
        # the same matrix up to the GPU matmul's rounding; align the DEVICE matrix on both paths
        lpz_h = lpz_d.cpu().numpy()
        tasks_d.append(on_dev.prepare_segmentation_task(text, lpz_d, f"u{i}", wav.shape[0]))
        tasks_h.append(on_host.prepare_segmentation_task(text, lpz_h, f"u{i}", wav.shape[0]))
    res_d = on_dev.get_segments_batch(tasks_d)
    res_h = on_host.get_segments_batch(tasks_h)
    checker = oracle_backed(pkg.CTCSegmentation(asr_cpu, kaldi_style_text=False, time_stamps="fixed", scoring_length=30), oracle)
    res_o = checker.get_segments_batch([checker.prepare_segmentation_task(t.text, t.lpz, t.name, None) for t in tasks_h])
    for a, b, o in zip(res_d, res_h, res_o):
        assert np.array_equal(a["timings"], o["timings"]) and np.array_equal(a["char_probs"], o["char_probs"])
        assert [s[:2] for s in a["segments"]] == [s[:2] for s in o["segments"]]
        np.testing.assert_allclose([s[2] for s in a["segments"]], [s[2] for s in o["segments"]], rtol=0, atol=1e-4)
        assert np.array_equal(a["timings"], b["timings"]) and np.array_equal(a["char_probs"], b["char_probs"])
        assert a["segments"] == b["segments"] and a["state_list"] == b["state_list"]


def test_batched_device_emissions_feed_the_dp(pkg, oracle):
    """get_lpz_batch on a GPU model with keep_lpz_on_device: one padded forward, the per-window
    matrices stay in HBM and go straight into one DP launch."""
    asr_gpu = FakeASR(seed=12, device="cuda:0")
    al = pkg.CTCSegmentation(asr_gpu, kaldi_style_text=False, time_stamps="fixed", scoring_length=30,
                             keep_lpz_on_device=True)
    host = pkg.CTCSegmentation(FakeASR(seed=12), kaldi_style_text=False, time_stamps="fixed", scoring_length=30)
    audio = NoiseAudio(40.0, 77)
    texts = [["HOLA QUE TAL", "MUY BIEN"], ["ADIOS AMIGOS"], ["UNO DOS TRES", "CUATRO", "CINCO SEIS"], ["SI"]]
    waves = []
    for i in range(len(texts)):
        clip, sr = audio.load(16000 * 4 * i, 16000 * (5 + 2 * i) + 37 * i)
        waves.append(asr_gpu.audio_normalizer(clip, sr))
    lpzs = al.get_lpz_batch(waves, frames_fn=pkg.wav2vec2_frames)
    assert all(z.is_cuda for z in lpzs)
    assert [z.shape[0] for z in lpzs] == [pkg.wav2vec2_frames(w.shape[0]) for w in waves]
    tasks_d = [al.prepare_segmentation_task(t, z, f"u{i}", w.shape[0]) for i, (t, z, w) in enumerate(zip(texts, lpzs, waves))]
    tasks_h = [host.prepare_segmentation_task(t, z.cpu().numpy(), f"u{i}", w.shape[0])
               for i, (t, z, w) in enumerate(zip(texts, lpzs, waves))]
    checker = oracle_backed(pkg.CTCSegmentation(FakeASR(seed=12), kaldi_style_text=False, time_stamps="fixed", scoring_length=30), oracle)
    res_o = checker.get_segments_batch([checker.prepare_segmentation_task(t.text, t.lpz, t.name, None) for t in tasks_h])
    for a, o in zip(al.get_segments_batch(tasks_d), res_o):   # the device-resident HIP path against the ORACLE
        assert np.array_equal(a["timings"], o["timings"]) and np.array_equal(a["char_probs"], o["char_probs"])
        assert [s[:2] for s in a["segments"]] == [s[:2] for s in o["segments"]]
        np.testing.assert_allclose([s[2] for s in a["segments"]], [s[2] for s in o["segments"]], rtol=0, atol=1e-4)


def test_benedetti_full_file_replay_hip_equals_oracle(pkg, oracle):
    """BASELINE.json configs[1]: all 157 rows of the reference's sample file (519 s) through the
    anchor iteration -- the HIP engine against the oracle answering the same requests, and the DP-call
    sequence against the committed fixture that ``bench.py --workload replay`` times."""
    import json
    import os

    from tests.replay_common import REPLAY_AUDIO_SECONDS, REPLAY_PARAMS, NoiseAudio as ReplayAudio, replay_vad
    gold = os.path.join(os.path.dirname(__file__), "golden")
    rows = json.load(open(os.path.join(gold, "benedetti_rows.json")))["rows"]
    recorded = json.load(open(os.path.join(gold, "replay_windows.json")))
    assert len(rows) == 157
    anchor = importlib.import_module(pkg.__name__ + ".anchor")
    asr = FakeASR(seed=5, sharp=6.0)

    def run(aligner):
        calls = []
        inner = aligner.get_segments_batch

        def recording(tasks, raise_errors=False):
            calls.extend((int(t.lpz.shape[0]), int(len(t.ground_truth_mat))) for t in tasks)
            return inner(tasks, raise_errors)
        aligner.get_segments_batch = recording
        co = anchor.file_alignment(asr, ReplayAudio(REPLAY_AUDIO_SECONDS, 2024), rows[0]["Sample_Path"],
                                   [dict(r) for r in rows], replay_vad(), 320.0, anchor.AnchorParams(**REPLAY_PARAMS))
        return anchor.run_batched([co], aligner)[0], calls

    mk = lambda: pkg.CTCSegmentation(asr, kaldi_style_text=False, time_stamps="fixed", scoring_length=30)
    hip, hip_calls = run(mk())
    ref, ref_calls = run(oracle_backed(mk(), oracle))
    assert hip_calls == ref_calls == [(c["T"], c["C"]) for c in recorded["calls"]]
    assert len(hip) == len(ref) == recorded["n_result_rows"]
    for a, b in zip(hip, ref):
        assert a[:6] == b[:6] and a[7:] == b[7:]
        assert abs(a[6] - b[6]) <= 1e-4 + 1e-9
