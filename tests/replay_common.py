"""Shared by the full-file replay (BASELINE.json configs[1]): its fixture generator, the GPU test
and bench.py's replay workload -- so that all three run the SAME window sequence."""
import torch

REPLAY_AUDIO_SECONDS = 540.0
# flags of align_utterances.sh:59-70 (threshold relaxed: the fake encoder on noise scores far lower
# than a real model on speech; what matters here is that accept / repeat / keep-previous all occur)
REPLAY_PARAMS = dict(threshold=-6.0, short_utterance_len=30, max_words_sequence=100, max_window_size=70.0,
                     window_to_stop=500.0, min_text_to_audio_prop=0.8, max_text_to_audio_prop_exec=10)


class NoiseAudio:
    """Deterministic pseudo-speech: seeded noise, so the fake encoder yields varied posteriors."""

    def __init__(self, seconds, seed, sr=16000):
        self.sample_rate, self.num_frames = sr, int(seconds * sr)
        g = torch.Generator().manual_seed(seed)
        self._data = torch.randn(self.num_frames, 1, generator=g) * 0.1

    def load(self, frame_offset, num_frames):
        if num_frames == 0 or num_frames < -1:   # torchaudio.load refuses these (the reference's bare except catches it)
            raise ValueError("Invalid argument: num_frames must be -1 or greater than 0.")
        end = None if num_frames == -1 else frame_offset + num_frames
        return self._data[frame_offset:end], self.sample_rate


def replay_vad(seconds=REPLAY_AUDIO_SECONDS):
    """Non-speech gaps >= 20 s removed, as filter_non_speech_segments.py --length 20 would leave them:
    three speech stretches with two long pauses."""
    cuts = [(0.0, 168.0), (191.0, 342.0), (365.0, seconds - 1.0)]
    return [dict(Start=s, End=e, Segment_Length=e - s) for s, e in cuts]
