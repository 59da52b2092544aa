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
