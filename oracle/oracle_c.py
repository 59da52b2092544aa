"""ORACLE -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.

ctypes binding of ``oracle/liboracle_ctcseg.so`` (built from
``ctc_segmentation_oracle.c`` by ``oracle/Makefile``).  See the C file's header for
what is restated, from where, and why parity against the real
``ctc-segmentation==1.7.1`` package is unpinned.

Importers allowed: tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg -- as the checker / reported baseline, never as the product.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_ctcseg.so")

OK, AUDIO_SHORTER_THAN_TEXT, INDEX_ERROR = 0, 1, 2


class OracleConfig(ctypes.Structure):
    _fields_ = [
        ("max_prob", ctypes.c_double),
        ("skip_prob", ctypes.c_double),
        ("min_window_size", ctypes.c_int32),
        ("max_window_size", ctypes.c_int32),
        ("index_duration", ctypes.c_double),
        ("score_min_mean_over_L", ctypes.c_int32),
        ("blank", ctypes.c_int32),
        ("blank_transition_cost_zero", ctypes.c_int32),
        ("preamble_transition_cost_zero", ctypes.c_int32),
        ("backtrack_from_max_t", ctypes.c_int32),
    ]


def build(force=False):
    src = os.path.join(_HERE, "ctc_segmentation_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_fill_table.restype = ctypes.c_int64
    return _lib


def make_config(**kw):
    cfg = OracleConfig()
    lib().oracle_default_config(ctypes.byref(cfg))
    for k, v in kw.items():
        if not hasattr(cfg, k):
            raise ValueError(k)
        setattr(cfg, k, v)
    return cfg


def _p(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def fill_table(lpz, gt, window, blank=0, flags=2):
    """-> (table [W,C] f32, offsets [C] i64, t_end)."""
    lpz = np.ascontiguousarray(lpz, np.float32)
    gt = np.ascontiguousarray(gt, np.int64)
    if gt.ndim == 1:
        gt = gt.reshape(-1, 1)
    T, V = lpz.shape
    C, S = gt.shape
    W = min(window, T)
    table = np.full((W, C), -1e10, np.float32)
    offsets = np.zeros(C, np.int64)
    t_end = lib().oracle_fill_table(
        _p(table, ctypes.c_float), ctypes.c_int64(W), ctypes.c_int64(C), _p(lpz, ctypes.c_float),
        ctypes.c_int64(T), ctypes.c_int64(V), _p(gt, ctypes.c_int64), ctypes.c_int64(S),
        _p(offsets, ctypes.c_int64), ctypes.c_int32(blank), ctypes.c_int32(flags))
    return table, offsets, int(t_end)


def get_segments(lpz, gt, utt_begin, cfg=None):
    """One segment end to end.  Returns a dict; ``status`` is OK / 1 / 2."""
    cfg = cfg or make_config()
    lpz = np.ascontiguousarray(lpz, np.float32)
    gt = np.ascontiguousarray(gt, np.int64)
    if gt.ndim == 1:
        gt = gt.reshape(-1, 1)
    utt_begin = np.ascontiguousarray(utt_begin, np.int64)
    T, V = lpz.shape
    C, S = gt.shape
    U = len(utt_begin) - 1
    timings = np.zeros(C)
    char_probs = np.zeros(T)
    state = np.zeros(T, np.int32)
    fol = np.zeros(C, np.int32)
    t_end = ctypes.c_int64(-1)
    seg = np.zeros((3, max(U, 1)))
    rc = lib().oracle_get_segments(
        ctypes.byref(cfg), _p(lpz, ctypes.c_float), ctypes.c_int64(T), ctypes.c_int64(V),
        _p(gt, ctypes.c_int64), ctypes.c_int64(C), ctypes.c_int64(S), _p(utt_begin, ctypes.c_int64),
        ctypes.c_int64(U), _p(timings, ctypes.c_double), _p(char_probs, ctypes.c_double),
        _p(state, ctypes.c_int32), _p(fol, ctypes.c_int32), ctypes.byref(t_end),
        _p(seg[0], ctypes.c_double), _p(seg[1], ctypes.c_double), _p(seg[2], ctypes.c_double))
    return dict(status=int(rc), timings=timings, char_probs=char_probs, state=state,
                frame_of_label=fol, t_end=int(t_end.value), seg_start=seg[0][:U], seg_end=seg[1][:U],
                seg_score=seg[2][:U])


def get_segments_batch(lpz_list, gt_list, utt_begin_list, cfg=None):
    """Loop of get_segments over a ragged batch (single thread)."""
    return [get_segments(l, g, u, cfg) for l, g, u in zip(lpz_list, gt_list, utt_begin_list)]


def time_uniform_batch(lpz, gt, utt_begin, cfg=None):
    """Time the C oracle on a uniform batch: lpz [B,T,V] f32, gt [B,C] i64, utt_begin [B,U+1] i64.

    One call into C for the whole batch (no Python loop inside the timed region).
    Returns (seconds, status[B]).  Used by bench.py's cpu_baseline leg (kind "port").
    """
    import time

    cfg = cfg or make_config()
    lpz = np.ascontiguousarray(lpz, np.float32)
    gt = np.ascontiguousarray(gt, np.int64)
    utt_begin = np.ascontiguousarray(utt_begin, np.int64)
    B, T, V = lpz.shape
    C = gt.shape[1]
    U = utt_begin.shape[1] - 1
    lpz_off = (np.arange(B, dtype=np.int64) * T * V)
    gt_off = (np.arange(B, dtype=np.int64) * C)
    utt_off = (np.arange(B, dtype=np.int64) * U)
    Ts = np.full(B, T, np.int32)
    Cs = np.full(B, C, np.int32)
    Us = np.full(B, U, np.int32)
    timings = np.zeros(B * C)
    char_probs = np.zeros(B * T)
    state = np.zeros(B * T, np.int32)
    fol = np.zeros(B * C, np.int32)
    t_end = np.zeros(B, np.int64)
    seg = np.zeros((3, B * U))
    status = np.zeros(B, np.int32)
    t0 = time.perf_counter()
    lib().oracle_get_segments_batch(
        ctypes.byref(cfg), ctypes.c_int64(B), _p(lpz, ctypes.c_float), _p(lpz_off, ctypes.c_int64),
        _p(Ts, ctypes.c_int32), ctypes.c_int64(V), _p(gt, ctypes.c_int64), _p(gt_off, ctypes.c_int64),
        _p(Cs, ctypes.c_int32), ctypes.c_int64(1), _p(utt_begin, ctypes.c_int64),
        _p(utt_off, ctypes.c_int64), _p(Us, ctypes.c_int32), _p(timings, ctypes.c_double),
        _p(char_probs, ctypes.c_double), _p(state, ctypes.c_int32), _p(fol, ctypes.c_int32),
        _p(t_end, ctypes.c_int64), _p(seg[0], ctypes.c_double), _p(seg[1], ctypes.c_double),
        _p(seg[2], ctypes.c_double), _p(status, ctypes.c_int32))
    dt = time.perf_counter() - t0
    return dt, status, dict(frame_of_label=fol.reshape(B, C), char_probs=char_probs.reshape(B, T),
                            t_end=t_end, seg_start=seg[0].reshape(B, U), seg_end=seg[1].reshape(B, U),
                            seg_score=seg[2].reshape(B, U))


def time_ragged_batch(segs, cfg=None):
    """Time the C oracle on a ragged batch ``[(lpz [T,V] f32, gt [C] i64, utt_begin [U+1] i64), ...]``:
    one call into C for the whole batch.  Returns (seconds, status[B]).  Used by the cpu_baseline leg."""
    import time

    cfg = cfg or make_config()
    B = len(segs)
    V = int(segs[0][0].shape[1])
    Ts = np.asarray([s[0].shape[0] for s in segs], np.int32)
    Cs = np.asarray([len(s[1]) for s in segs], np.int32)
    Us = np.asarray([len(s[2]) - 1 for s in segs], np.int32)
    lpz = np.ascontiguousarray(np.concatenate([np.asarray(s[0], np.float32).reshape(-1) for s in segs]))
    gt = np.ascontiguousarray(np.concatenate([np.asarray(s[1], np.int64).reshape(-1) for s in segs]))
    ub = np.ascontiguousarray(np.concatenate([np.asarray(s[2], np.int64).reshape(-1) for s in segs]))
    lpz_off = np.concatenate([[0], np.cumsum(Ts.astype(np.int64) * V)])[:-1].astype(np.int64)
    gt_off = np.concatenate([[0], np.cumsum(Cs.astype(np.int64))])[:-1].astype(np.int64)
    utt_off = np.concatenate([[0], np.cumsum(Us.astype(np.int64))])[:-1].astype(np.int64)
    nT, nC, nU = int(Ts.sum()), int(Cs.sum()), max(1, int(Us.sum()))
    timings = np.zeros(nC)
    char_probs = np.zeros(nT)
    state = np.zeros(nT, np.int32)
    fol = np.zeros(nC, np.int32)
    t_end = np.zeros(B, np.int64)
    seg = np.zeros((3, nU))
    status = np.zeros(B, np.int32)
    t0 = time.perf_counter()
    lib().oracle_get_segments_batch(
        ctypes.byref(cfg), ctypes.c_int64(B), _p(lpz, ctypes.c_float), _p(lpz_off, ctypes.c_int64),
        _p(Ts, ctypes.c_int32), ctypes.c_int64(V), _p(gt, ctypes.c_int64), _p(gt_off, ctypes.c_int64),
        _p(Cs, ctypes.c_int32), ctypes.c_int64(1), _p(ub, ctypes.c_int64),
        _p(utt_off, ctypes.c_int64), _p(Us, ctypes.c_int32), _p(timings, ctypes.c_double),
        _p(char_probs, ctypes.c_double), _p(state, ctypes.c_int32), _p(fol, ctypes.c_int32),
        _p(t_end, ctypes.c_int64), _p(seg[0], ctypes.c_double), _p(seg[1], ctypes.c_double),
        _p(seg[2], ctypes.c_double), _p(status, ctypes.c_int32))
    return time.perf_counter() - t0, status
