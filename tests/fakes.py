"""Test doubles for the acoustic side of the aligner protocol (no SpeechBrain, no audio files).

FakeASR quacks like the ``EncoderASR`` object the reference builds at
src/iterative_utterance_alignment.py:415: ``tokenizer``, ``encode_batch``,
``hparams.log_softmax``, ``audio_normalizer``, ``device``.  Its encoder is a fixed random
projection of 25 ms frames at a 20 ms hop (the wav2vec2 geometry: T = (N - 400) // 320 + 1).
"""
from types import SimpleNamespace

import numpy as np
import torch


class FakeTokenizer:
    """Character-level SentencePiece look-alike: 32 pieces, id 0 = blank, id 1 = <unk>."""

    def __init__(self):
        letters = list("ABCDEFGHIJKLMNOPQRSTUVWXYZÑÁÉ")
        self.pieces = ["<blank>", "<unk>", "▁"] + letters
        assert len(self.pieces) == 32
        self._id = {p: i for i, p in enumerate(self.pieces)}

    def vocab_size(self):
        return len(self.pieces)

    def id_to_piece(self, i):
        return self.pieces[i]

    def unk_id(self):
        return 1

    def encode_as_pieces(self, text):
        out = ["▁"]
        for ch in text:
            out.append("▁" if ch == " " else (ch if ch in self._id else "<unk>"))
        return out

    def encode_as_ids(self, text):
        return [self._id[p] for p in self.encode_as_pieces(text)]


class FakeASR:
    def __init__(self, seed=0, device="cpu", vocab=32, sharp=4.0):
        self.device = device
        self.tokenizer = FakeTokenizer()
        g = torch.Generator().manual_seed(seed)
        self._proj = (torch.randn(400, vocab, generator=g) * sharp / 20.0).to(device)
        self.hparams = SimpleNamespace(log_softmax=torch.nn.LogSoftmax(dim=-1), sample_rate=16000)
        self.mods = SimpleNamespace()

    def audio_normalizer(self, audio, sr):
        if audio.dim() == 2:
            audio = audio.mean(dim=1)
        return audio

    def encode_batch(self, wavs, wav_lens):
        frames = wavs.unfold(1, 400, 320)           # [B, T, 400]
        return frames @ self._proj.to(wavs.device)  # [B, T, V]


# ---------------------------------------------------------------------------------------
# Scripted aligner: drives the anchor state machine without any acoustic model.  Scores are
# a pure function of (utterance text, clip length), so that the reference's own loop
# (tests/golden/make_anchor_goldens.py) and this repo's loop see identical "alignments".
# ---------------------------------------------------------------------------------------
import zlib


class ScriptedTask:
    def __init__(self, text, name, n_frames, speech_len):
        self.text, self.name, self.n_frames, self.speech_len = text, name, n_frames, speech_len
        self.segments = None

    def set(self, **kw):
        self.__dict__.update(kw)

    def __str__(self):
        out = ""
        for i, seg in enumerate(self.segments):
            out += f"{self.name}_{i:04} {self.name} {seg[0]:.2f} {seg[1]:.2f} {seg[2]:3.4f} {self.text[i]}\n"
        return out


class ScriptedAligner:
    """Quacks like CTCSegmentation for the caller loops (get_lpz / prepare_segmentation_task /
    get_segments).  mode: 'mixed' pseudo-random scores in [-3.5, 0], 'good' all -0.2,
    'bad' all -3.0, 'lastbad' only the last utterance of a task is bad, 'improving*' scores
    that get better as utterances are dropped."""

    def __init__(self, mode="mixed", salt=0, fs=16000):
        self.mode, self.salt, self.fs = mode, salt, fs
        self.calls = []

    def estimate_samples_to_frames_ratio(self):
        return 320.0

    def get_lpz(self, audio):
        n = int(audio.shape[0])
        return np.zeros((max(1, n // 320), 4), np.float32)

    def prepare_segmentation_task(self, text, lpz, name=None, speech_len=None):
        if isinstance(text, str):
            text = text.splitlines()
        text = [t for t in text if len(t)]
        return ScriptedTask(text, name, lpz.shape[0], speech_len)

    def _score(self, utt, task, i):
        if self.mode == "good":
            return -0.2
        if self.mode == "bad":
            return -3.0
        if self.mode == "lastbad":
            return -3.0 if i == len(task.text) - 1 else -0.3
        if self.mode == "improving":      # fewer utterances in the window -> better scores
            return -0.5 - 0.3 * len(task.text)
        if self.mode == "improving_fast":
            return -0.2 - 0.45 * len(task.text)
        if self.mode == "improving_slow":
            return -1.05 - 0.1 * len(task.text) - 0.01 * (zlib.crc32(utt.encode()) % 7)
        h = zlib.crc32(f"{self.salt}|{utt}|{task.speech_len}".encode()) % 1000
        return -3.5 * h / 1000.0

    def get_segments(self, task):
        chars = sum(len(t) for t in task.text) + len(task.text) + 1
        self.calls.append((task.name, len(task.text), task.speech_len))
        if chars > task.n_frames:
            raise AssertionError("Audio is shorter than text!")
        dur = task.speech_len / self.fs
        total = float(sum(len(t) for t in task.text)) or 1.0
        segs, acc = [], 0.0
        for i, utt in enumerate(task.text):
            d = 0.9 * dur * len(utt) / total
            segs.append((acc + 0.01, acc + d, self._score(utt, task, i)))
            acc += d
        return {"segments": segs}


class ScriptedASR:
    """asr_model stand-in: only audio_normalizer is used by the caller loops."""

    def audio_normalizer(self, audio, sr):
        return audio[:, 0] if audio.ndim == 2 else audio


def oracle_backed(aligner, oracle):
    """Test-only: make a product CTCSegmentation answer get_segments / get_segments_batch from
    the CPU oracle instead of the HIP engine (same protocol, same result dict)."""
    import numpy as _np

    def get_segments(task):
        cfg = task.config
        ocfg = oracle.make_config(index_duration=cfg.index_duration_in_seconds, blank=cfg.blank,
                                  score_min_mean_over_L=cfg.score_min_mean_over_L,
                                  min_window_size=cfg.min_window_size, max_window_size=cfg.max_window_size,
                                  preamble_transition_cost_zero=int(cfg.preamble_transition_cost_zero),
                                  backtrack_from_max_t=int(cfg.backtrack_from_max_t))
        r = oracle.get_segments(task.lpz, task.ground_truth_mat[:, 0], _np.asarray(task.utt_begin_indices), ocfg)
        if r["status"] == 1:
            raise AssertionError("Audio is shorter than text!")
        if r["status"] == 2:
            raise IndexError("backtrack left the trellis")
        return {"name": task.name, "timings": r["timings"], "char_probs": r["char_probs"], "state_list": None,
                "segments": list(zip(r["seg_start"].tolist(), r["seg_end"].tolist(), r["seg_score"].tolist())),
                "done": True}

    def get_segments_batch(tasks, raise_errors=False):
        out = []
        for t in tasks:
            try:
                out.append(get_segments(t))
            except (AssertionError, IndexError) as e:
                out.append(e)
        return out

    aligner.get_segments = get_segments
    aligner.get_segments_batch = get_segments_batch
    return aligner
