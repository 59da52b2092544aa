"""Host mirror of the ``ctc_segmentation`` package surface the reference depends on.

The reference never imports ``ctc_segmentation`` directly; it reaches it through
SpeechBrain's wrapper (``speechbrain.alignment.ctc_segmentation``, imported at
/root/reference/src/iterative_utterance_alignment.py:11, word_level_alignment.py:8,
search_on_speech.py:9) and configures it through the parameter names listed at
/root/reference/src/test/test_ctc_segmentation.py:20-38.  This module keeps those
names and argument meanings (``ctc-segmentation==1.7.1``, requirements.txt:13):

* ``CtcSegmentationParameters``   -- same fields and defaults;
* ``prepare_token_list`` / ``prepare_text`` -- text -> label matrix (host work, negligible);
* ``ctc_segmentation(config, lpz, ground_truth)`` -- table fill + backtrack, executed by the
  HIP kernels behind the C ABI (``include/ctcfa.h``); raises ``AssertionError`` when the
  text is longer than the audio, as the package does (caught by the reference at
  iterative_utterance_alignment.py:390);
* ``get_segments_device`` -- fill + backtrack + ``determine_utterance_segments`` in one
  launch for a batch of segments.

There is no CPU implementation of the DP here: without the HIP library and a GPU these
functions raise.
"""
import numpy as np

from . import _native


class CtcSegmentationParameters:
    """Parameters of the segmentation (names/defaults of ctc-segmentation 1.7.1)."""

    max_prob = -10000000000.0
    skip_prob = -10000000000.0
    min_window_size = 8000
    max_window_size = 100000
    index_duration = 0.025
    score_min_mean_over_L = 30
    space = "·"
    blank = 0
    replace_spaces_with_blanks = False
    blank_transition_cost_zero = False
    preamble_transition_cost_zero = True
    backtrack_from_max_t = False
    self_transition = "ε"
    start_of_ground_truth = "#"
    excluded_characters = ".,»«•❍·"
    tokenized_meta_symbol = "▁"
    char_list = None
    subsampling_factor = None
    frame_duration_ms = None

    def __init__(self, **kwargs):
        self.set(**kwargs)

    def set(self, **kwargs):
        for key, value in kwargs.items():
            if not hasattr(self, key):
                raise ValueError(f"Parameter {key} is not a CtcSegmentationParameters field")
            setattr(self, key, value)

    @property
    def index_duration_in_seconds(self):
        if self.subsampling_factor and self.frame_duration_ms:
            return self.frame_duration_ms * self.subsampling_factor / 1000
        return self.index_duration

    @property
    def flags(self):
        return int(self.blank_transition_cost_zero) + 2 * int(self.preamble_transition_cost_zero)

    def __repr__(self):
        fields = ("min_window_size", "max_window_size", "index_duration", "score_min_mean_over_L",
                  "blank", "blank_transition_cost_zero", "preamble_transition_cost_zero",
                  "backtrack_from_max_t")
        return "CtcSegmentationParameters(" + ", ".join(f"{k}={getattr(self, k)!r}" for k in fields) + ")"

    def to_native(self):
        """-> ``ctcfa_params`` for the C ABI."""
        if self.max_prob != -10000000000.0 or self.skip_prob > self.max_prob:
            raise NotImplementedError("max_prob / skip_prob other than the package defaults")
        flags = self.flags
        if self.backtrack_from_max_t:
            flags |= _native.FLAG_BACKTRACK_FROM_MAX_T
        key = (int(self.blank), flags, int(self.min_window_size), int(self.max_window_size),
               int(self.score_min_mean_over_L), float(self.index_duration_in_seconds))
        cached = self.__dict__.get("_native_params")
        if cached is None or cached[0] != key:   # (one struct per parameter set, not per call)
            cached = (key, _native.default_params(blank=key[0], flags=key[1], min_window_size=key[2], max_window_size=key[3],
                                                  score_min_mean_over_L=key[4], index_duration=key[5]))
            self.__dict__["_native_params"] = cached
        return cached[1]


def prepare_token_list(config, text):
    """Token-id utterances -> (ground_truth_mat int64 [C,1], utt_begin_indices).

    ``text`` is a list of 1-D integer arrays.  Layout: ``[-1]``, then for every
    utterance a separating ``blank`` (unless one is already there) followed by its ids,
    and a closing ``blank``.  This is the converter SpeechBrain uses for
    ``text_converter="tokenize"`` (its default; the reference scripts never override it).
    """
    blank = config.blank
    seq = [-1]
    begins = []
    for utt in text:
        if seq[-1] != blank:
            seq.append(blank)
        begins.append(len(seq) - 1)
        seq.extend(int(i) for i in np.asarray(utt).reshape(-1))
    if seq[-1] != blank:
        seq.append(blank)
    begins.append(len(seq) - 1)
    return np.asarray(seq, dtype=np.int64).reshape(-1, 1), begins


def prepare_text(config, text, char_list=None):
    """Character utterances -> (ground_truth_mat int64 [C,S], utt_begin_indices).

    ``text_converter="classic"``: a ``"#"``-led character string with ``config.space``
    between utterances; row i of the matrix holds, for s = 0..S-1, the index of the
    ``char_list`` entry equal to the s+1 characters ending at position i (or -1).
    """
    if isinstance(config.blank, str):
        config.blank = 0
    if char_list is not None:
        config.char_list = char_list
    chars = config.char_list
    blank_sym = chars[config.blank]
    gt = config.start_of_ground_truth
    begins = []
    for utt in text:
        if not gt.endswith(config.space):
            gt += config.space
        begins.append(len(gt) - 1)
        for ch in utt:
            if ch.isspace() and config.replace_spaces_with_blanks:
                if not gt.endswith(config.space):
                    gt += config.space
            elif ch in chars and ch not in config.excluded_characters:
                gt += ch
            elif config.tokenized_meta_symbol + ch in chars:
                gt += ch
    if not gt.endswith(config.space):
        gt += config.space
    begins.append(len(gt) - 1)
    span_max = max(len(c) for c in chars)
    index_of = {}
    for i, c in enumerate(chars):
        index_of.setdefault(c, i)  # list.index semantics: first match
    mat = np.full((len(gt), span_max), -1, dtype=np.int64)
    for i in range(len(gt)):
        for s in range(span_max):
            if i - s < 0:
                continue
            span = gt[i - s:i + 1].replace(config.space, blank_sym)
            j = index_of.get(span)
            if j is not None:
                mat[i, s] = j
    return mat, begins


_engines = {}


def default_engine(device=0):
    """Process-wide engine for ``device`` (created on first use; needs a GPU)."""
    eng = _engines.get(device)
    if eng is None:
        eng = _engines[device] = _native.Engine(device)
    return eng


def _labels_from_mat(ground_truth):
    """ground_truth_mat -> the label sequence int32 [C] -- or, when tokens of more than one character
    occur in it (columns 1.. not all -1: the "classic" text converter), the matrix int32 [C, S]."""
    gt = np.asarray(ground_truth)
    if gt.ndim == 2:
        if gt.shape[1] != 1 and (gt[:, 1:] != -1).any():
            used = 1 + int(np.nonzero((gt[:, 1:] != -1).any(axis=0))[0].max())
            return np.ascontiguousarray(gt[:, :used + 1], dtype=np.int32)
        gt = gt[:, 0]
    return np.ascontiguousarray(gt, dtype=np.int32)


MAX_LABEL_WIDTH = 16   # ctcfa::kMaxSpan


def _align_label_matrices(engine, config, lpz_list, labels, utt_begin_list, want_state):
    """Segments whose ground truth holds multi-character tokens: ``ctcfa_align_batch_spans`` (the
    literal, sequential kernel)."""
    S = max(g.shape[1] if g.ndim == 2 else 1 for g in labels)
    if S > MAX_LABEL_WIDTH:
        raise NotImplementedError(f"tokens of more than {MAX_LABEL_WIDTH} characters")
    V = int(lpz_list[0].shape[1])
    mats = []
    for b, g in enumerate(labels):
        m = np.full((len(g), S), -1, np.int32)
        if g.ndim == 2:
            m[:, :g.shape[1]] = g
        else:
            m[:, 0] = g
        if len(m) < 2 or (m[0] != -1).any():
            raise ValueError(f"segment {b}: ground truth must start with a row of -1 and hold at least one more")
        if int(m.min()) < -1 or int(m.max()) >= V:
            raise IndexError(f"segment {b}: label id outside the vocabulary [0, {V})")
        mats.append(m)
    # (emission shapes and utterance starts: the checks of the single-label path, on a stand-in sequence of the same length)
    _validate_segments(lpz_list, [np.r_[-1, np.zeros(len(m) - 1, np.int32)].astype(np.int32) for m in mats], utt_begin_list)
    lpz_list = [np.ascontiguousarray(l.cpu().numpy() if hasattr(l, "cpu") else l, dtype=np.float32) for l in lpz_list]
    return engine.align_batch(config.to_native(), lpz_list, mats, utt_begin_list, want_state=want_state, label_width=S)


def _raise_for_status(status, error=None):
    if error is not None:
        raise error
    if status == _native.ST_AUDIO_SHORTER_THAN_TEXT:
        raise AssertionError("Audio is shorter than text!")
    if status == _native.ST_BACKTRACK_FAILED:
        raise IndexError("CTC segmentation backtrack left the trellis")
    if status == _native.ST_WINDOWED_UNSUPPORTED:
        raise NotImplementedError("windowed DP regime with more than ~40 000 frames (one column must fit the LDS)")
    if status == _native.ST_TEXT_TOO_LONG:
        raise NotImplementedError("more label columns than one workgroup of the fill kernel covers "
                                  "(ctcfa_max_label_columns: 5 485 for a 32-entry vocabulary, 5 119 for the others up to 256)")
    if status == _native.ST_TOO_MANY_LABELS:   # (only plans created with texts_of_31_labels: this mirror looks at the labels itself)
        raise ValueError("the text uses more than 31 vocabulary entries beside the blank in a plan that promised otherwise")
    if status != _native.ST_OK:
        raise RuntimeError(f"ctcfa status {status}")


def _is_device_tensor(x):
    return hasattr(x, "is_cuda") and bool(x.is_cuda)


ST_INVALID_INPUT = -1   # (Python side only) the segment's inputs were refused on the host: the result dict carries "error"


def _segment_errors(lpz_list, labels, utt_begin_list):
    """Per segment: the exception NumPy would have answered with in the package (label ids outside the
    vocabulary, utterance starts outside the label sequence, ...) or None -- the kernels index emissions with
    the raw label, so such a segment never reaches them -- and whether its label sequence holds a -1 after the
    first entry (a column without a token: the package skips it; served by the label-matrix kernel)."""
    errors, minus_one = [None] * len(lpz_list), [False] * len(lpz_list)
    if not lpz_list:
        return errors, minus_one
    # the launch's vocabulary width: the one most well-formed segments agree on (ties: the earliest) -- a malformed or
    # odd first segment is that segment's error, not the launch's
    widths = [int(l.shape[1]) for l in lpz_list if getattr(l, "ndim", 0) == 2]
    V = max(dict.fromkeys(widths), key=widths.count) if widths else 0
    for b, (l, g) in enumerate(zip(lpz_list, labels)):
        if getattr(l, "ndim", 0) != 2 or int(l.shape[1]) != V:
            errors[b] = ValueError(f"segment {b}: emissions must be [T, {V}] like the launch's other segments, got {tuple(getattr(l, 'shape', ()))}")
        elif len(g) < 2 or g[0] != -1:
            errors[b] = ValueError(f"segment {b}: ground truth must start with -1 and hold at least one more label")
        else:
            inner = g[1:]
            minus_one[b] = bool((inner == -1).any())
            # (one pass: as unsigned numbers, negative ids are far above any vocabulary)
            rest = inner[inner != -1] if minus_one[b] else inner
            if len(rest) and int(rest.view(np.uint32).max()) >= V:
                errors[b] = IndexError(f"segment {b}: label id outside the vocabulary [0, {V})")
    if utt_begin_list is not None:
        for b, (u, g) in enumerate(zip(utt_begin_list, labels)):
            if errors[b] is not None:
                continue
            if getattr(u, "ndim", 1) != 1 or len(u) < 1:
                errors[b] = ValueError(f"segment {b}: utt_begin_indices must hold U + 1 indices")
                continue
            ul = u.tolist() if hasattr(u, "tolist") else list(u)
            C = len(g)
            # determine_utterance_segments reads timings[i-1], timings[i], timings[i+1] for every start i
            # and timings[e-1], timings[e] for the closing index e
            if len(ul) > 1 and (min(ul[:-1]) < 1 or max(ul[:-1]) > C - 2 or ul[-1] < 1 or ul[-1] > C - 1):
                errors[b] = IndexError(f"segment {b}: utterance start outside the label sequence (C = {C})")
    return errors, minus_one


def _validate_segments(lpz_list, labels, utt_begin_list):
    """Raise the first of ``_segment_errors`` (callers that hand over ONE launch's worth of trusted shapes)."""
    for e in _segment_errors(lpz_list, labels, utt_begin_list)[0]:
        if e is not None:
            raise e


def _refused(error, T, C, U):
    """Result dict of a segment that was refused on the host (the other segments of the call are aligned)."""
    return {"status": ST_INVALID_INPUT, "error": error, "t_end": -1, "frame_of_label": np.zeros(C, np.int32),
            "char_prob": np.zeros(T, np.float32), "state": np.full(T, -2, np.int32),
            "seg_start": np.zeros(U), "seg_end": np.zeros(U), "seg_score": np.zeros(U)}


def get_segments_device(config, lpz_list, ground_truth_list, utt_begin_list, engine=None, want_state=True):
    """Batch of segments through the HIP engine -> list of per-segment dicts.

    Each dict: status, t_end, frame_of_label (int32 [C]), char_prob (fp32 [T]),
    state (int32 [T]: label id | -1 self transition | -2 untouched),
    seg_start / seg_end / seg_score (fp64 [U]).  No exception for per-segment status.

    ``lpz_list`` entries are host arrays (the reference's protocol: ``get_lpz`` returns
    NumPy) or CUDA ``torch`` tensors; in the second case the emissions never leave HBM
    (``ctcfa_plan_run_device``), only the small result vectors come back to the host.
    """
    engine = engine or default_engine()
    labels = [_labels_from_mat(g) for g in ground_truth_list]
    if any(g.ndim == 2 for g in labels):
        return _align_label_matrices(engine, config, lpz_list, labels, utt_begin_list, want_state)
    errors, minus_one = _segment_errors(lpz_list, labels, utt_begin_list)
    if any(e is not None for e in errors) or any(minus_one):
        # one refused segment is that segment's error, not the launch's; a sequence with a -1 inside goes through
        # the label-matrix kernel (as a [C, 1] matrix), the rest as usual
        out = [None] * len(lpz_list)
        for b, e in enumerate(errors):
            if e is not None:
                U = 0 if utt_begin_list is None else max(0, len(utt_begin_list[b]) - 1)
                out[b] = _refused(e, int(lpz_list[b].shape[0]) if getattr(lpz_list[b], "ndim", 0) >= 1 else 0, len(labels[b]), U)
        for pick in ([b for b in range(len(out)) if out[b] is None and not minus_one[b]],
                     [b for b in range(len(out)) if out[b] is None and minus_one[b]]):
            if not pick:
                continue
            sub_gt = [labels[b].reshape(-1, 1) if minus_one[b] else labels[b] for b in pick]
            sub_ub = None if utt_begin_list is None else [utt_begin_list[b] for b in pick]
            if minus_one[pick[0]]:
                part = _align_label_matrices(engine, config, [lpz_list[b] for b in pick], sub_gt, sub_ub, want_state)
            else:
                part = get_segments_device(config, [lpz_list[b] for b in pick], sub_gt, sub_ub, engine=engine, want_state=want_state)
            for b, r in zip(pick, part):
                out[b] = r
        return out
    emission_of = shared_emissions(lpz_list)
    if lpz_list and all(_is_device_tensor(l) for l in lpz_list):
        return _align_resident(engine, config, lpz_list, labels, utt_begin_list, want_state, emission_of)
    own = {}
    for b, l in enumerate(lpz_list):   # (one conversion per emission block, not per segment)
        e = b if emission_of is None else emission_of[b]
        if e not in own:
            own[e] = np.ascontiguousarray(l.cpu().numpy() if hasattr(l, "cpu") else l, dtype=np.float32)
    lpz_list = [own[b if emission_of is None else emission_of[b]] for b in range(len(lpz_list))]
    return engine.align_batch(config.to_native(), lpz_list, labels, utt_begin_list, want_state=want_state,
                              emission_of=emission_of)


def shared_emissions(lpz_list):
    """``emission_of`` for ``ctcfa_align_batch_shared``: entries of ``lpz_list`` that are THE SAME OBJECT
    (the anchor loop's repeats: one ``lpz`` per window, the text shrinking by its last utterance,
    ``iterative_utterance_alignment.py:201-219``) share their emissions -- and, where one text is a
    prefix of the other, the trellis fill.  None when nothing is shared."""
    first = {}
    out = []
    for b, l in enumerate(lpz_list):
        out.append(first.setdefault(id(l), b))
    return out if any(e != b for b, e in enumerate(out)) else None


def _align_resident(engine, config, lpz_list, labels, utt_begin_list, want_state, emission_of=None):
    """Device-resident emissions (torch CUDA tensors): ``ctcfa_align_batch_resident`` on the current
    torch stream -- the emissions are used where they are, the small inputs go up in one packed copy,
    the results come back in one; no per-call device allocation."""
    import torch

    dev = lpz_list[0].device
    shapes = [(int(l.shape[0]), int(l.shape[1])) for l in lpz_list]
    flat = [l.reshape(-1) if l.dtype == torch.float32 and l.is_contiguous() else l.to(torch.float32).contiguous().reshape(-1)
            for b, l in enumerate(lpz_list) if emission_of is None or emission_of[b] == b]
    d_lpz = flat[0] if len(flat) == 1 else torch.cat(flat)   # (one window: no copy at all)
    return engine.align_batch(config.to_native(), None, labels, utt_begin_list, want_state=want_state,
                              d_lpz=d_lpz.data_ptr(), stream=torch.cuda.current_stream(dev).cuda_stream, shapes=shapes,
                              emission_of=emission_of)


def ctc_segmentation(config, lpz, ground_truth, engine=None):
    """Fill + backtrack for one segment -> (timings fp64 [C], char_probs fp64 [T], state_list)."""
    res = get_segments_device(config, [lpz], [ground_truth], None, engine=engine)[0]
    _raise_for_status(res["status"], res.get("error"))
    timings = res["frame_of_label"].astype(np.int64) * config.index_duration_in_seconds
    return timings, res["char_prob"].astype(np.float64), state_list_from(config, res["state"])


def state_list_from(config, state):
    """int32 state codes -> the package's ``state_list`` (label symbol, "ε" or "")."""
    chars = config.char_list
    out = []
    for s in np.asarray(state).tolist():
        if s == -2:
            out.append("")
        elif s == -1:
            out.append(config.self_transition)
        else:
            out.append(chars[s] if chars is not None else s)
    return out
