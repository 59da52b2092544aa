"""ORACLE -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.

NumPy/pure-Python twin of ``oracle/ctc_segmentation_oracle.c``, laid out like the
un-vendored package ``ctc-segmentation==1.7.1`` (/root/reference/requirements.txt:13;
modules ``ctc_segmentation.py`` and ``ctc_segmentation_dyn.pyx``) so that anyone who
has the real package can diff the two function by function.  The package cannot be
installed here (no network) and the reference's tests hold no expected values
(src/test/test_ctc_segmentation.py:40-43), so this twin is pinned by nothing but
the C restatement next to it: PARITY UNPINNED.

Pure-Python loops: use it for small cases only (T*C up to ~1e5 cells).
Only tests/ may import this module; the product path never does.

Reference call sites that fix the observable protocol:
  src/iterative_utterance_alignment.py:208-219, src/word_level_alignment.py:92-103,
  src/search_on_speech.py:77-88.
"""
import math

import numpy as np

f32 = np.float32


class CtcSegmentationParameters:
    """Defaults of ctc_segmentation.CtcSegmentationParameters (1.7.1)."""

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
        for key in kwargs:
            if not hasattr(self, key) and key != "index_duration":
                raise ValueError(f"unknown parameter {key}")
            setattr(self, key, kwargs[key])

    @property
    def index_duration_in_seconds(self):
        if self.subsampling_factor and self.frame_duration_ms:
            return self.frame_duration_ms * self.subsampling_factor / 1000
        return self.index_duration

    @property
    def flags(self):
        return int(self.blank_transition_cost_zero) + 2 * int(self.preamble_transition_cost_zero)


def _cmax(a, b):
    """Cython/Python max(a, b): the later operand wins only if strictly greater."""
    return b if b > a else a


def cython_fill_table(table, lpz, ground_truth, offsets, blank, flags):
    """ctc_segmentation_dyn.pyx::cython_fill_table, fp32 locals, wraparound indexing."""
    W, C = table.shape
    T = lpz.shape[0]
    S = ground_truth.shape[1]
    prob_max = f32(-1000000000)
    offset = 0
    offset_sum = 0
    cur_offset = np.zeros([S], np.int64) - 1
    mean_offset = f32((T - W) / float(C))
    higher_offset = int(mean_offset) + 1
    last_arg_max = -1
    last_max = f32(0)
    table[0, 0] = 0
    for c in range(C):
        if c > 0:
            offset = min(max(0, last_arg_max - W // 2), min(higher_offset, (T - W) - offset_sum))
            for s in range(S - 1):
                cur_offset[s + 1] = cur_offset[s] + offset
            cur_offset[0] = offset
            offset_sum += offset
        offsets[c] = offset_sum
        last_arg_max = -1
        last_max = f32(0)
        for t in range(1 if c == 0 else 0, W):
            switch_prob = prob_max
            max_lpz_prob = prob_max
            for s in range(S):
                if ground_truth[c, s] != -1:
                    e = lpz[t + offset_sum, ground_truth[c, s]]
                    if t >= W - (cur_offset[s] - 1) or t - 1 + cur_offset[s] < 0:
                        p = prob_max
                    else:
                        p = f32(table[t - 1 + cur_offset[s], c - (s + 1)] + e)
                    switch_prob = _cmax(switch_prob, p)
                    max_lpz_prob = _cmax(max_lpz_prob, e)
            if t - 1 < 0:
                stay_prob = prob_max
            elif c == 0 and (flags & 2):
                stay_prob = f32(0)
            else:
                stay_step = _cmax(lpz[t + offset_sum, blank], max_lpz_prob)
                if (flags & 1) and ground_truth[c, 0] == blank:
                    stay_step = f32(0)  # SURVEY A.6 U3 (recalled semantics)
                stay_prob = f32(table[t - 1, c] + stay_step)
            table[t, c] = _cmax(switch_prob, stay_prob)
            if last_arg_max == -1 or last_max < table[t, c]:
                last_max = table[t, c]
                last_arg_max = t
    return last_arg_max, C - 1


def _residual(prob, prob_is_f32, est32):
    """abs(prob - est) with NumPy-1.24 scalar typing: fp32 when both are np.float32."""
    if prob_is_f32:
        return float(abs(f32(f32(prob) - est32)))
    return abs(float(prob) - float(est32))


def ctc_segmentation(config, lpz, ground_truth):
    """ctc_segmentation.py::ctc_segmentation -> (timings, char_probs, state_list)."""
    blank = config.blank
    offset = 0
    lpz = np.ascontiguousarray(lpz, dtype=np.float32)
    ground_truth = np.asarray(ground_truth, dtype=np.int64)
    if len(ground_truth) > lpz.shape[0] and config.skip_prob <= config.max_prob:
        raise AssertionError("Audio is shorter than text!")
    window_size = config.min_window_size
    while True:
        table = np.zeros([min(window_size, lpz.shape[0]), len(ground_truth)], dtype=np.float32)
        table.fill(config.max_prob)
        offsets = np.zeros([len(ground_truth)], dtype=np.int64)
        t, c = cython_fill_table(table, lpz, ground_truth, offsets, config.blank, config.flags)
        if config.backtrack_from_max_t:
            t = table.shape[0] - 1
        timings = np.zeros([len(ground_truth)])
        char_probs = np.zeros([lpz.shape[0]])
        state_list = [""] * lpz.shape[0]
        try:
            while t != 0 or c != 0:
                min_s = None
                min_switch_prob_delta = np.inf
                max_lpz_prob, max_is_f32 = config.max_prob, False
                for s in range(ground_truth.shape[1]):
                    if ground_truth[c, s] != -1:
                        offset = offsets[c] - (offsets[c - 1 - s] if c - s > 0 else 0)
                        if c > 0:
                            switch_prob, sp_f32 = lpz[t + offsets[c], ground_truth[c, s]], True
                        else:
                            switch_prob, sp_f32 = config.max_prob, False
                        est_switch_prob = f32(table[t, c] - table[t - 1 + offset, c - 1 - s])
                        delta = _residual(switch_prob, sp_f32, est_switch_prob)
                        if delta < min_switch_prob_delta:
                            min_switch_prob_delta = delta
                            min_s = s
                        if switch_prob > max_lpz_prob:
                            max_lpz_prob, max_is_f32 = switch_prob, sp_f32
                if t > 0:
                    lb = lpz[t + offsets[c], blank]
                    if max_lpz_prob > lb:
                        stay_prob, st_f32 = max_lpz_prob, max_is_f32
                    else:
                        stay_prob, st_f32 = lb, True
                else:
                    stay_prob, st_f32 = config.max_prob, False
                est_stay_prob = f32(table[t, c] - table[t - 1, c])
                if _residual(stay_prob, st_f32, est_stay_prob) > min_switch_prob_delta:
                    if c > 0:
                        for s in range(0, min_s + 1):
                            timings[c - s] = (offsets[c] + t) * config.index_duration_in_seconds
                        char_probs[offsets[c] + t] = max_lpz_prob
                        char_index = ground_truth[c, min_s]
                        state_list[offsets[c] + t] = (
                            config.char_list[char_index] if config.char_list else int(char_index)
                        )
                    c -= 1 + min_s
                    t -= 1 - offset
                else:
                    char_probs[offsets[c] + t] = stay_prob
                    state_list[offsets[c] + t] = config.self_transition
                    t -= 1
        except IndexError:
            window_size *= 2
            if window_size < config.max_window_size:
                continue
            raise
        break
    return timings, char_probs, state_list


def determine_utterance_segments(config, utt_begin_indices, char_probs, timings, text):
    """ctc_segmentation.py::determine_utterance_segments -> [(start, end, min_avg)]."""

    def compute_time(index, align_type):
        middle = (timings[index] + timings[index - 1]) / 2
        if align_type == "begin":
            return max(timings[index + 1] - 0.5, middle)
        return min(timings[index - 1] + 0.5, middle)

    segments = []
    min_prob = np.float64(-10000000000.0)
    for i in range(len(text)):
        start = compute_time(utt_begin_indices[i], "begin")
        end = compute_time(utt_begin_indices[i + 1], "end")
        start_t = int(round(start / config.index_duration_in_seconds))
        end_t = int(round(end / config.index_duration_in_seconds))
        n = config.score_min_mean_over_L
        if end_t <= start_t:
            min_avg = min_prob
        elif end_t - start_t <= n:
            min_avg = char_probs[start_t:end_t].mean()
        else:
            min_avg = np.float64(0.0)
            for t in range(start_t, end_t - n):
                min_avg = min(min_avg, char_probs[t : t + n].mean())
        segments.append((start, end, min_avg))
    return segments


def prepare_token_list(config, text):
    """ctc_segmentation.py::prepare_token_list; ``text`` = list of 1-D int arrays."""
    ground_truth = [-1]
    utt_begin_indices = []
    for utt in text:
        if not ground_truth[-1] == config.blank:
            ground_truth += [config.blank]
        utt_begin_indices.append(len(ground_truth) - 1)
        ground_truth += np.asarray(utt).tolist()
    if not ground_truth[-1] == config.blank:
        ground_truth += [config.blank]
    utt_begin_indices.append(len(ground_truth) - 1)
    ground_truth_mat = np.array(ground_truth, dtype=np.int64).reshape(-1, 1)
    return ground_truth_mat, utt_begin_indices


def prepare_text(config, text, char_list=None):
    """ctc_segmentation.py::prepare_text (text_converter="classic")."""
    if type(config.blank) == str:
        config.blank = 0
    if char_list is not None:
        config.char_list = char_list
    blank = config.char_list[config.blank]
    ground_truth = config.start_of_ground_truth
    utt_begin_indices = []
    for utt in text:
        if not ground_truth.endswith(config.space):
            ground_truth += config.space
        utt_begin_indices.append(len(ground_truth) - 1)
        for char in utt:
            if char.isspace() and config.replace_spaces_with_blanks:
                if not ground_truth.endswith(config.space):
                    ground_truth += config.space
            elif char in config.char_list and char not in config.excluded_characters:
                ground_truth += char
            elif config.tokenized_meta_symbol + char in config.char_list:
                ground_truth += char
    if not ground_truth.endswith(config.space):
        ground_truth += config.space
    utt_begin_indices.append(len(ground_truth) - 1)
    max_char_len = max([len(c) for c in config.char_list])
    ground_truth_mat = np.ones([len(ground_truth), max_char_len], np.int64) * -1
    for i in range(len(ground_truth)):
        for s in range(max_char_len):
            if i - s < 0:
                continue
            span = ground_truth[i - s : i + 1]
            span = span.replace(config.space, blank)
            if span in config.char_list:
                ground_truth_mat[i, s] = config.char_list.index(span)
    return ground_truth_mat, utt_begin_indices
