"""Aligner protocol of the reference: ``CTCSegmentation`` / ``CTCSegmentationTask``.

Drop-in for ``speechbrain.alignment.ctc_segmentation`` (speechbrain==0.5.11,
/root/reference/requirements.txt:87) as the reference uses it:

    aligner = CTCSegmentation(asr_model, kaldi_style_text=False, time_stamps="fixed", scoring_length=30)
                                                        iterative_utterance_alignment.py:419
    ratio   = aligner.estimate_samples_to_frames_ratio()                             :420
    lpz     = aligner.get_lpz(audio_normalized)                                      :201
    task    = aligner.prepare_segmentation_task(text, lpz, name, speech_len)         :208-213
    segments = aligner.get_segments(task); task.set(**segments); str(task)           :216-218

(also word_level_alignment.py:26,89-102 and search_on_speech.py:36,74-87; keyword
arguments and ``aligner.config.*`` fields as in src/test/test_ctc_segmentation.py:20-38.)

``get_lpz`` stays PyTorch host code (the acoustic model forward, on ROCm when the model
lives on a GPU); ``get_segments`` runs the HIP DP engine.  ``get_segments_batch`` is the
addition that lets the anchor-iteration / word-level callers put many windows in one launch.
"""
import copy
from pathlib import Path
from types import SimpleNamespace

import numpy as np

from . import ctc_segmentation as cs
from .ctc_segmentation import CtcSegmentationParameters

try:  # torch is only needed for the acoustic-model side (get_lpz)
    import torch
except Exception:  # pragma: no cover
    torch = None


def wav2vec2_frames(n_samples):
    """Frames the wav2vec2 / HuBERT convolutional front end yields for n samples (receptive
    field 400, hop 320): the ``frames_fn`` of ``CTCSegmentation.get_lpz_batch``."""
    return (int(n_samples) - 400) // 320 + 1


class CTCSegmentationTask(SimpleNamespace):
    """Task object: inputs and results of one segmentation (``str(task)`` prints segments)."""

    text = None
    ground_truth_mat = None
    utt_begin_indices = None
    timings = None
    char_probs = None
    state_list = None
    segments = None
    config = None
    done = False
    name = "utt"
    utt_ids = None
    lpz = None
    print_confidence_score = True
    print_utterance_text = True

    def set(self, **kwargs):
        self.__dict__.update(kwargs)

    def __str__(self):
        lines = []
        n = len(self.segments)
        if self.utt_ids is None:
            names = [f"{self.name}_{i:04}" for i in range(n)]
        else:
            assert n == len(self.utt_ids)
            names = self.utt_ids
        for i, seg in enumerate(self.segments):
            line = f"{names[i]} {self.name} {seg[0]:.2f} {seg[1]:.2f}"
            if self.print_confidence_score:
                line += f" {seg[2]:3.4f}"
            if self.print_utterance_text:
                line += f" {self.text[i]}"
            lines.append(line + "\n")
        return "".join(lines)


class CTCSegmentation:
    """Align text to audio with CTC segmentation (MI355X engine behind SpeechBrain's API)."""

    fs = 16000
    kaldi_style_text = True
    samples_to_frames_ratio = None
    time_stamps = "auto"
    choices_time_stamps = ["auto", "fixed"]
    text_converter = "tokenize"
    choices_text_converter = ["tokenize", "classic"]
    warned_about_misconfiguration = False

    def __init__(self, asr_model, kaldi_style_text=True, text_converter="tokenize",
                 time_stamps="auto", engine=None, keep_lpz_on_device=False, **ctc_segmentation_args):
        if not hasattr(asr_model, "tokenizer"):
            raise AttributeError("The ASR model needs a tokenizer (asr_model.tokenizer)")
        self.config = CtcSegmentationParameters()
        self.asr_model = asr_model
        self._engine = engine
        # False: get_lpz returns a host NumPy array (the reference's protocol).  True: when the
        # model runs on a GPU the log-posteriors stay in HBM as a torch tensor and feed the DP
        # kernels without a PCIe round trip.
        self.keep_lpz_on_device = bool(keep_lpz_on_device)
        self._encode = asr_model.encode_batch
        mods = getattr(asr_model, "mods", None)
        decoder = getattr(mods, "decoder", None) if mods is not None else None
        if decoder is not None and hasattr(decoder, "ctc_forward_step"):
            self._ctc = decoder.ctc_forward_step  # encoder-decoder model: log-softmax included
        else:
            self._ctc = asr_model.hparams.log_softmax
        self._tokenizer = asr_model.tokenizer
        self.set_config(fs=getattr(getattr(asr_model, "hparams", None), "sample_rate", 16000),
                        time_stamps=time_stamps, kaldi_style_text=kaldi_style_text,
                        text_converter=text_converter, **ctc_segmentation_args)
        tok = self._tokenizer
        self.config.char_list = [tok.id_to_piece(i) for i in range(tok.vocab_size())]

    # -- configuration -----------------------------------------------------------------
    def set_config(self, time_stamps=None, fs=None, samples_to_frames_ratio=None, set_blank=None,
                   replace_spaces_with_blanks=None, kaldi_style_text=None, text_converter=None,
                   gratis_blank=None, min_window_size=None, max_window_size=None,
                   scoring_length=None):
        if time_stamps is not None:
            if time_stamps not in self.choices_time_stamps:
                raise NotImplementedError(f"Parameter time_stamps has to be one of {self.choices_time_stamps}")
            self.time_stamps = time_stamps
        if fs is not None:
            self.fs = float(fs)
        if samples_to_frames_ratio is not None:
            self.samples_to_frames_ratio = float(samples_to_frames_ratio)
        if set_blank is not None:
            self.config.blank = int(set_blank)
        if replace_spaces_with_blanks is not None:
            self.config.replace_spaces_with_blanks = bool(replace_spaces_with_blanks)
        if kaldi_style_text is not None:
            self.kaldi_style_text = bool(kaldi_style_text)
        if text_converter is not None:
            if text_converter not in self.choices_text_converter:
                raise NotImplementedError(f"Parameter text_converter has to be one of {self.choices_text_converter}")
            self.text_converter = text_converter
        if min_window_size is not None:
            self.config.min_window_size = int(min_window_size)
        if max_window_size is not None:
            self.config.max_window_size = int(max_window_size)
        if gratis_blank is not None:
            self.config.blank_transition_cost_zero = bool(gratis_blank)
        if scoring_length is not None:
            self.config.score_min_mean_over_L = int(scoring_length)

    def get_timing_config(self, speech_len=None, lpz_len=None):
        timing_cfg = {"index_duration": self.config.index_duration}
        if self.time_stamps == "fixed":
            if self.samples_to_frames_ratio is None:
                self.samples_to_frames_ratio = self.estimate_samples_to_frames_ratio()
            index_duration = self.samples_to_frames_ratio / self.fs
        else:
            assert self.time_stamps == "auto"
            index_duration = (speech_len / lpz_len) / self.fs
        timing_cfg["index_duration"] = index_duration
        return timing_cfg

    def estimate_samples_to_frames_ratio(self, speech_len=215040):
        random_input = torch.rand(speech_len)
        lpz = self.get_lpz(random_input)
        return speech_len / lpz.shape[0]

    # -- acoustic model ----------------------------------------------------------------
    def get_lpz(self, speech):
        """[N] waveform -> [T, V] fp32 log-posteriors (host NumPy array, as in the reference)."""
        with torch.no_grad():
            if isinstance(speech, np.ndarray):
                speech = torch.tensor(speech)
            device = getattr(self.asr_model, "device", "cpu")
            speech = speech.unsqueeze(0).to(device)
            wav_lens = torch.tensor([1.0]).to(device)
            enc = self._encode(speech, wav_lens)
            lpz = self._ctc(enc).detach().squeeze(0)
            if self.keep_lpz_on_device and lpz.is_cuda:
                return lpz.to(torch.float32).contiguous()
            return lpz.cpu().numpy()

    def get_lpz_batch(self, speeches, frames_fn=None):
        """Many waveforms -> list of [T_i, V] log-posteriors from ONE padded encoder forward
        (SURVEY.md §8f N1).  ``wav_lens`` carries the relative lengths, SpeechBrain's convention
        for padded batches.  ``frames_fn(n_samples) -> T_i`` says how many leading frames belong
        to an item (``wav2vec2_frames`` for the wav2vec2/HuBERT feature extractor); the default
        is SpeechBrain's own ``round(relative_length * T_max)``.
        Not the reference's numerics for models with self-attention: the reference encodes every
        window alone (``wav_lens=[1.0]``, src/test/test_asr.py:35), a padded batch lets the
        model's length mask decide what the padding contributes.  Convolution-only encoders give
        identical frames.  Callers opt in (``anchor.run_batched(..., batch_lpz=True)``)."""
        with torch.no_grad():
            waves = [torch.as_tensor(s) for s in speeches]
            lens = [int(w.shape[0]) for w in waves]
            n_max = max(lens)
            device = getattr(self.asr_model, "device", "cpu")
            batch = torch.zeros(len(waves), n_max, dtype=waves[0].dtype)
            for i, w in enumerate(waves):
                batch[i, : lens[i]] = w
            wav_lens = torch.tensor([n / n_max for n in lens], dtype=torch.float32)
            enc = self._encode(batch.to(device), wav_lens.to(device))
            lpz = self._ctc(enc).detach()
            t_max = lpz.shape[1]
            out = []
            for i, n in enumerate(lens):
                t_i = int(frames_fn(n)) if frames_fn is not None else int(round(n / n_max * t_max))
                t_i = max(1, min(t_max, t_i))
                item = lpz[i, :t_i]
                if self.keep_lpz_on_device and item.is_cuda:
                    out.append(item.to(torch.float32).contiguous())
                else:
                    out.append(item.cpu().numpy())
            return out

    # -- text ----------------------------------------------------------------------------
    def _split_text(self, text):
        utt_ids = None
        if isinstance(text, str):
            text = text.splitlines()
        text = list(filter(len, text))
        if self.kaldi_style_text:
            pairs = [utt.split(" ", 1) for utt in text]
            pairs = [p for p in pairs if len(p) == 2]
            utt_ids = [p[0] for p in pairs]
            text = [p[1] for p in pairs]
        return utt_ids, text

    def prepare_segmentation_task(self, text, lpz, name=None, speech_len=None):
        if self.time_stamps == "auto" and speech_len is None:
            raise ValueError("speech_len is needed for time_stamps='auto'")
        self.config.set(**self.get_timing_config(speech_len, lpz.shape[0]))
        # every task keeps its OWN copy: with time_stamps="auto" the index duration differs from task
        # to task, and a batch of tasks prepared up front must not all see the last one's value
        config = copy.copy(self.config)
        utt_ids, text = self._split_text(text)
        if self.text_converter == "tokenize":
            tok = self._tokenizer
            unk = tok.unk_id()
            token_list = [np.array(tok.encode_as_ids(utt)) for utt in text]
            token_list = [utt[utt != unk] for utt in token_list]
            ground_truth_mat, utt_begin_indices = cs.prepare_token_list(config, token_list)
        else:
            assert self.text_converter == "classic"
            pieces = [" ".join(self._tokenizer.encode_as_pieces(utt)) for utt in text]
            pieces = [utt.replace("<unk>", "") for utt in pieces]
            ground_truth_mat, utt_begin_indices = cs.prepare_text(config, pieces)
        return CTCSegmentationTask(
            config=config, name=name, text=text, ground_truth_mat=ground_truth_mat,
            utt_begin_indices=utt_begin_indices, utt_ids=utt_ids, timings=None, lpz=lpz, done=False)

    # -- the hot path --------------------------------------------------------------------
    def _engine_or_default(self):
        return self._engine or cs.default_engine()

    @staticmethod
    def _result_dict(task, res):
        cs._raise_for_status(res["status"], res.get("error"))
        config = task.config
        timings = res["frame_of_label"].astype(np.int64) * config.index_duration_in_seconds
        segments = list(zip(res["seg_start"].tolist(), res["seg_end"].tolist(), res["seg_score"].tolist()))
        return {"name": task.name, "timings": timings, "char_probs": res["char_prob"].astype(np.float64),
                "state_list": cs.state_list_from(config, res["state"]), "segments": segments, "done": True}

    def get_segments(self, task):
        """One task -> result dict to be splatted into ``task.set(**segments)``."""
        assert isinstance(task, CTCSegmentationTask)
        assert task.config is not None
        res = cs.get_segments_device(task.config, [task.lpz], [task.ground_truth_mat],
                                     [task.utt_begin_indices], engine=self._engine_or_default())[0]
        return self._result_dict(task, res)

    def get_segments_batch(self, tasks, raise_errors=False):
        """Many tasks in one launch.  Returns a list of result dicts; a task whose
        status is not OK yields the exception instance the reference would have seen
        (``AssertionError`` for text longer than audio) instead of a dict."""
        if not tasks:
            return []
        # one launch per distinct parameter set (time_stamps="auto": the index duration is per task)
        groups = {}
        for i, t in enumerate(tasks):
            groups.setdefault(repr(t.config), []).append(i)
        res = [None] * len(tasks)
        for idx in groups.values():
            part = cs.get_segments_device(tasks[idx[0]].config, [tasks[i].lpz for i in idx],
                                          [tasks[i].ground_truth_mat for i in idx],
                                          [tasks[i].utt_begin_indices for i in idx], engine=self._engine_or_default())
            for i, r in zip(idx, part):
                res[i] = r
        out = []
        for task, r in zip(tasks, res):
            try:
                out.append(self._result_dict(task, r))
            except (AssertionError, IndexError, NotImplementedError, ValueError) as e:
                if raise_errors:
                    raise
                out.append(e)
        return out

    def __call__(self, speech_or_path, text, name=None):
        if isinstance(speech_or_path, (str, Path)):
            speech = self.asr_model.load_audio(speech_or_path)
        else:
            speech = speech_or_path
        lpz = self.get_lpz(speech)
        task = self.prepare_segmentation_task(text, lpz, name, speech.shape[0])
        task.set(**self.get_segments(task))
        return task
