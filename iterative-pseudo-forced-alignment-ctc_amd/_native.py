"""ctypes binding of the C ABI in ``include/ctcfa.h`` (``csrc/libctcfa_hip.so``).

The shared library is the product's only compute path for the CTC-segmentation DP
(the part the reference reaches through ``aligner.get_segments(task)``,
/root/reference/src/iterative_utterance_alignment.py:216).  There is no CPU
fallback: if the library is missing or no HIP device is present every call here
raises.  Build it with ``python -c "import __graft_entry__ as g; g.build()"``.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CTCFA_LIB: alternative build of the same library (kernel-tuning experiments only)
LIB_PATH = os.environ.get("CTCFA_LIB") or os.path.join(_HERE, "csrc", "libctcfa_hip.so")

OK, ERR_INVALID, ERR_HIP, ERR_UNSUPPORTED, ERR_NOMEM = 0, 1, 2, 3, 4
ST_OK, ST_AUDIO_SHORTER_THAN_TEXT, ST_BACKTRACK_FAILED, ST_WINDOWED_UNSUPPORTED, ST_TEXT_TOO_LONG, ST_INTERNAL, ST_TOO_MANY_LABELS = 0, 1, 2, 3, 4, 5, 6
FLAG_BLANK_TRANSITION_COST_ZERO, FLAG_PREAMBLE_TRANSITION_COST_ZERO, FLAG_BACKTRACK_FROM_MAX_T, FLAG_TEXTS_OF_31_LABELS = 1, 2, 4, 8

# every symbol include/ctcfa.h declares
EXPORTS = (
    "ctcfa_version", "ctcfa_status_string", "ctcfa_engine_create", "ctcfa_engine_destroy",
    "ctcfa_last_error", "ctcfa_default_params", "ctcfa_plan_create", "ctcfa_plan_destroy",
    "ctcfa_plan_get_info", "ctcfa_plan_run_device", "ctcfa_plan_run_pipelined", "ctcfa_plan_flush",
    "ctcfa_plan_get_timings", "ctcfa_plan_get_step_intervals",
    "ctcfa_plan_set_timing", "ctcfa_plan_set_timing_stride", "ctcfa_align_batch", "ctcfa_align_batch_resident",
    "ctcfa_align_batch_shared", "ctcfa_plan_create_shared", "ctcfa_plan_get_sharing", "ctcfa_align_batch_spans",
    "ctcfa_build_flags", "ctcfa_max_label_columns",
)


class NativeLibraryError(RuntimeError):
    pass


class Params(ctypes.Structure):
    _fields_ = [
        ("blank", ctypes.c_int32),
        ("flags", ctypes.c_uint32),
        ("min_window_size", ctypes.c_int32),
        ("max_window_size", ctypes.c_int32),
        ("score_min_mean_over_L", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
        ("index_duration", ctypes.c_double),
    ]


class PlanInfo(ctypes.Structure):
    _fields_ = [
        ("batch", ctypes.c_int32),
        ("cols_per_lane", ctypes.c_int32),
        ("waves_per_seg", ctypes.c_int32),
        ("vocab_pitch", ctypes.c_int32),
        ("lds_bytes", ctypes.c_int32),
        ("n_blocks_max", ctypes.c_int32),
        ("workspace_bytes", ctypes.c_int64),
        ("algorithmic_bytes", ctypes.c_int64),
        ("total_frames", ctypes.c_int64),
    ]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


_lib = None


def load():
    """Load the HIP library; raise loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} not found: the HIP extension is not built and there is no CPU fallback. "
            "Run __graft_entry__.build().")
    lib = ctypes.CDLL(LIB_PATH)
    lib.ctcfa_build_flags.restype = ctypes.c_int
    flags = lib.ctcfa_build_flags()
    if flags and os.environ.get("CTCFA_ALLOW_TUNING_BUILD") != "1":
        raise NativeLibraryError(
            f"{LIB_PATH} is a kernel-tuning build (ctcfa_build_flags() = {flags}: 1 ablated -- WRONG results, 2 cycle "
            "stamps in output buffers, 4 one vocabulary pitch, 8 retuned): not a library to align with.  "
            "Set CTCFA_ALLOW_TUNING_BUILD=1 for the tuning tools.")
    vp = ctypes.c_void_p
    i32p = ctypes.POINTER(ctypes.c_int32)
    lib.ctcfa_version.restype = ctypes.c_int
    lib.ctcfa_status_string.restype = ctypes.c_char_p
    lib.ctcfa_status_string.argtypes = [ctypes.c_int]
    lib.ctcfa_last_error.restype = ctypes.c_char_p
    lib.ctcfa_last_error.argtypes = [vp]
    lib.ctcfa_max_label_columns.argtypes = [vp, ctypes.c_int32]
    lib.ctcfa_max_label_columns.restype = ctypes.c_int
    lib.ctcfa_engine_create.argtypes = [ctypes.POINTER(vp), ctypes.c_int]
    lib.ctcfa_engine_destroy.argtypes = [vp]
    lib.ctcfa_engine_destroy.restype = None
    lib.ctcfa_default_params.argtypes = [ctypes.POINTER(Params)]
    lib.ctcfa_default_params.restype = None
    lib.ctcfa_plan_create.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(Params), ctypes.c_int32,
                                      ctypes.c_int32, i32p, i32p, i32p, ctypes.c_int32]
    lib.ctcfa_plan_destroy.argtypes = [vp]
    lib.ctcfa_plan_destroy.restype = None
    lib.ctcfa_plan_get_info.argtypes = [vp, ctypes.POINTER(PlanInfo)]
    lib.ctcfa_plan_run_device.argtypes = [vp] + [vp] * 12
    lib.ctcfa_plan_run_pipelined.argtypes = [vp] + [vp] * 12
    lib.ctcfa_plan_flush.argtypes = [vp, vp]
    lib.ctcfa_plan_get_timings.argtypes = [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_float),
                                           ctypes.POINTER(ctypes.c_float)]
    lib.ctcfa_plan_get_step_intervals.argtypes = [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
    lib.ctcfa_plan_set_timing.argtypes = [vp, ctypes.c_int]
    lib.ctcfa_plan_set_timing_stride.argtypes = [vp, ctypes.c_int]
    lib.ctcfa_align_batch.argtypes = [vp, ctypes.POINTER(Params), ctypes.c_int32, ctypes.c_int32,
                                      i32p, i32p, i32p] + [vp] * 11
    lib.ctcfa_align_batch_resident.argtypes = [vp, ctypes.POINTER(Params), ctypes.c_int32, ctypes.c_int32,
                                               i32p, i32p, i32p] + [vp] * 12
    lib.ctcfa_align_batch_shared.argtypes = [vp, ctypes.POINTER(Params), ctypes.c_int32, ctypes.c_int32,
                                             i32p, i32p, i32p, i32p, vp, ctypes.c_int32] + [vp] * 11
    lib.ctcfa_plan_create_shared.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(Params), ctypes.c_int32,
                                             ctypes.c_int32, i32p, i32p, i32p, i32p, i32p, ctypes.c_int32]
    lib.ctcfa_plan_get_sharing.argtypes = [vp, i32p, i32p]
    lib.ctcfa_align_batch_spans.argtypes = [vp, ctypes.POINTER(Params), ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                            i32p, i32p, i32p] + [vp] * 11
    _lib = lib
    return lib


def default_params(**kw):
    p = Params()
    load().ctcfa_default_params(ctypes.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise ValueError(f"unknown parameter {k}")
        setattr(p, k, v)
    return p


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _i32p(a):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


class Engine:
    """One per (process, device): wraps ``ctcfa_engine`` (owns a HIP stream)."""

    def __init__(self, device=0):
        self._lib = load()
        h = ctypes.c_void_p()
        rc = self._lib.ctcfa_engine_create(ctypes.byref(h), int(device))
        if rc != OK:
            msg = self._lib.ctcfa_last_error(None).decode()
            raise NativeLibraryError(f"ctcfa_engine_create failed ({rc}): {msg}")
        self._h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ctcfa_engine_destroy(self._h)
            self._h = None

    def max_label_columns(self, vocab):
        """Label columns the widest fill launch shape covers for this vocabulary (more: status 4)."""
        return int(self._lib.ctcfa_max_label_columns(self._h, int(vocab)))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != OK:
            msg = self._lib.ctcfa_last_error(self._h).decode()
            exc = NotImplementedError if rc == ERR_UNSUPPORTED else (ValueError if rc == ERR_INVALID else NativeLibraryError)
            raise exc(f"{what} failed ({rc}): {msg}")

    # Host-side staging of the host-buffer entries: grow-only NumPy buffers whose addresses are turned into
    # ctypes pointers once (``ndarray.ctypes`` costs microseconds per use -- with a dozen arguments per call that
    # was as long as the kernels of a 10 s window).  Results are copied out of them, so a call's result
    # arrays stay valid after the next call.
    def _staging(self, B, nT, nC, nU, nL, nB):
        st = getattr(self, "_st", None)
        if st is None or st["B"] < B or st["T"] < nT or st["C"] < nC or st["U"] < nU or st["L"] < nL or st["UB"] < nB:
            grow = lambda need, old: max(need + need // 2 + 16, old)
            o = st or dict(B=0, T=0, C=0, U=0, L=0, UB=0)
            cap = dict(B=grow(B, o["B"]), T=grow(nT, o["T"]), C=grow(nC, o["C"]), U=grow(nU, o["U"]), L=grow(nL, o["L"]),
                       UB=grow(nB, o["UB"]))
            st = dict(cap)
            st["Tn"] = np.zeros(cap["B"], np.int32)
            st["Cn"] = np.zeros(cap["B"], np.int32)
            st["Un"] = np.zeros(cap["B"], np.int32)
            st["em"] = np.zeros(cap["B"], np.int32)
            st["lab"] = np.zeros(cap["L"], np.int32)
            st["ub"] = np.zeros(cap["UB"], np.int32)
            st["fol"] = np.zeros(cap["C"], np.int32)
            st["cp"] = np.zeros(cap["T"], np.float32)
            st["state"] = np.zeros(cap["T"], np.int32)
            st["seg"] = np.zeros((3, cap["U"]), np.float64)
            st["t_end"] = np.zeros(cap["B"], np.int32)
            st["status"] = np.zeros(cap["B"], np.int32)
            i32 = ctypes.POINTER(ctypes.c_int32)
            st["p"] = {k: st[k].ctypes.data_as(i32) for k in ("Tn", "Cn", "Un", "em")}
            st["p"].update({k: st[k].ctypes.data_as(ctypes.c_void_p) for k in ("lab", "ub", "fol", "cp", "state", "t_end", "status")})
            st["p"]["seg"] = [st["seg"][r].ctypes.data_as(ctypes.c_void_p) for r in range(3)]
            self._st = st
        return st

    def align_batch(self, params, lpz_list, labels_list, utt_begin_list=None, want_state=True, d_lpz=None,
                    stream=None, shapes=None, emission_of=None, label_width=1):
        """Host-buffer entry ``ctcfa_align_batch``, or -- with ``d_lpz`` (device address of the
        concatenated fp32 emissions) and ``shapes`` = [(T_b, V), ...] -- ``ctcfa_align_batch_resident``.

        lpz_list: fp32 [T_b, V]; labels_list: int [C_b] (ground_truth_mat[:, 0]);
        utt_begin_list: int [U_b + 1] or None.  Returns a list of per-segment dicts.

        emission_of (``ctcfa_align_batch_shared``): emission_of[b] = index of the segment whose
        emissions segment b uses; only the blocks of segments with emission_of[b] == b are read from
        ``lpz_list`` / expected at ``d_lpz`` (``shapes`` still lists every segment).

        label_width = S > 1 (``ctcfa_align_batch_spans``, host emissions only): every entry of
        ``labels_list`` is a label matrix int [C_b, S] (ground_truth_mat, -1 padded).
        """
        shapes = shapes or [l.shape for l in lpz_list]
        B = len(shapes)
        V = int(shapes[0][1])
        S = int(label_width)
        if S > 1 and (d_lpz is not None or emission_of is not None):
            raise ValueError("label matrices take host emissions of their own")
        have_utt = utt_begin_list is not None
        Ts = [int(sh[0]) for sh in shapes]
        Cs = [len(g) for g in labels_list]
        Us = [len(u) - 1 for u in utt_begin_list] if have_utt else None
        nT, nC = sum(Ts), sum(Cs)
        nU = sum(Us) if have_utt else 0
        if emission_of is not None:
            if len(emission_of) != B:
                raise ValueError("emission_of must have one entry per segment")
            if all(int(e) == b for b, e in enumerate(emission_of)):
                emission_of = None
        st = self._staging(B, nT, nC, max(nU, 1), nC * S, nU + B)
        p = st["p"]
        st["Tn"][:B] = Ts
        st["Cn"][:B] = Cs
        if have_utt:
            st["Un"][:B] = Us
        if emission_of is not None:
            st["em"][:B] = emission_of
        o = 0
        for g in labels_list:
            n = len(g) * S
            st["lab"][o:o + n] = np.asarray(g).reshape(-1)
            o += n
        if have_utt:
            o = 0
            for u in utt_begin_list:
                st["ub"][o:o + len(u)] = u
                o += len(u)
        if d_lpz is None:
            own = [l for b, l in enumerate(lpz_list) if emission_of is None or int(emission_of[b]) == b]
            lpz = own[0] if len(own) == 1 else np.concatenate([np.asarray(l, np.float32).reshape(-1) for l in own])
            lpz = np.ascontiguousarray(lpz, dtype=np.float32)
            lpz_arg = ctypes.c_void_p(lpz.ctypes.data)
        else:
            lpz_arg = ctypes.c_void_p(int(d_lpz))
        tail = (p["lab"], p["ub"] if have_utt else None, p["fol"], p["cp"], p["state"] if want_state else None,
                p["seg"][0] if have_utt else None, p["seg"][1] if have_utt else None, p["seg"][2] if have_utt else None,
                p["t_end"], p["status"])
        pU = p["Un"] if have_utt else None
        byref = ctypes.byref(params)
        if S > 1:
            rc = self._lib.ctcfa_align_batch_spans(self._h, byref, B, V, S, p["Tn"], p["Cn"], pU, lpz_arg, *tail)
            self._check(rc, "ctcfa_align_batch_spans")
        elif emission_of is not None:
            rc = self._lib.ctcfa_align_batch_shared(self._h, byref, B, V, p["Tn"], p["Cn"], pU, p["em"], lpz_arg,
                                                    0 if d_lpz is None else 1, *tail,
                                                    ctypes.c_void_p(stream) if stream else None)
            self._check(rc, "ctcfa_align_batch_shared")
        elif d_lpz is None:
            rc = self._lib.ctcfa_align_batch(self._h, byref, B, V, p["Tn"], p["Cn"], pU, lpz_arg, *tail)
            self._check(rc, "ctcfa_align_batch")
        else:
            rc = self._lib.ctcfa_align_batch_resident(self._h, byref, B, V, p["Tn"], p["Cn"], pU, lpz_arg, *tail,
                                                      ctypes.c_void_p(stream) if stream else None)
            self._check(rc, "ctcfa_align_batch_resident")
        fol, cp, state, seg = st["fol"], st["cp"], st["state"], st["seg"]
        status, t_end = st["status"][:B].tolist(), st["t_end"][:B].tolist()
        out = []
        t0 = c0 = u0 = 0
        for b in range(B):
            t1, c1 = t0 + Ts[b], c0 + Cs[b]
            d = dict(status=status[b], t_end=t_end[b], frame_of_label=fol[c0:c1].copy(), char_prob=cp[t0:t1].copy())
            if want_state:
                d["state"] = state[t0:t1].copy()
            if have_utt:
                u1 = u0 + Us[b]
                d["seg_start"] = seg[0, u0:u1].copy()
                d["seg_end"] = seg[1, u0:u1].copy()
                d["seg_score"] = seg[2, u0:u1].copy()
                u0 = u1
            t0, c0 = t1, c1
            out.append(d)
        return out

    def plan(self, params, vocab, T, C, U=None, force_cols_per_lane=0, emission_of=None, labels=None, texts_of_31_labels=False):
        return Plan(self, params, vocab, T, C, U, force_cols_per_lane, emission_of, labels, texts_of_31_labels)


class Plan:
    """Batch geometry + HBM workspace (``ctcfa_plan``); run it on device-resident buffers."""

    def __init__(self, engine, params, vocab, T, C, U=None, force_cols_per_lane=0, emission_of=None, labels=None,
                 texts_of_31_labels=False):
        """emission_of / labels: ``ctcfa_plan_create_shared`` (segments that share emissions; ``labels`` =
        all segments' labels back to back, for the prefix check and for narrowing, or None).
        texts_of_31_labels: CTCFA_FLAG_TEXTS_OF_31_LABELS -- the caller's promise that no text uses more than 31
        vocabulary entries beside the blank (a narrowed plan for vocabularies of 33 .. 256 entries)."""
        self._eng = engine
        self._lib = engine._lib
        T, C = _i32(T), _i32(C)
        U = _i32(U) if U is not None else None
        h = ctypes.c_void_p()
        if texts_of_31_labels:
            params = type(params).from_buffer_copy(params)
            params.flags |= FLAG_TEXTS_OF_31_LABELS
        if emission_of is None and labels is not None:
            # labels alone: nothing shared, but the plan may look at the texts -- a vocabulary above 32 entries whose
            # segments use at most 31 labels each gets a NARROWED plan
            emission_of = np.arange(len(T), dtype=np.int32)
        if emission_of is None:
            rc = self._lib.ctcfa_plan_create(engine._h, ctypes.byref(h), ctypes.byref(params), len(T), int(vocab),
                                             _i32p(T), _i32p(C), _i32p(U), int(force_cols_per_lane))
            engine._check(rc, "ctcfa_plan_create")
        else:
            emission_of = _i32(emission_of)
            labels = _i32(labels) if labels is not None else None
            if len(emission_of) != len(T) or (labels is not None and len(labels) != int(C.sum())):
                raise ValueError("emission_of: one entry per segment; labels: sum(C) entries")
            rc = self._lib.ctcfa_plan_create_shared(engine._h, ctypes.byref(h), ctypes.byref(params), len(T), int(vocab),
                                                    _i32p(T), _i32p(C), _i32p(U), _i32p(emission_of), _i32p(labels),
                                                    int(force_cols_per_lane))
            engine._check(rc, "ctcfa_plan_create_shared")
        self._h = h
        self.T, self.C, self.U = T, C, U
        info = PlanInfo()
        self._lib.ctcfa_plan_get_info(h, ctypes.byref(info))
        self.info = info.as_dict()

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ctcfa_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sharing(self):
        """-> (trellis fills, emission blocks) this batch needs."""
        a, b = ctypes.c_int32(), ctypes.c_int32()
        self._eng._check(self._lib.ctcfa_plan_get_sharing(self._h, ctypes.byref(a), ctypes.byref(b)), "ctcfa_plan_get_sharing")
        return a.value, b.value

    def set_timing_stride(self, stride):
        """Record timing events only on every ``stride``-th run."""
        self._eng._check(self._lib.ctcfa_plan_set_timing_stride(self._h, int(stride)), "ctcfa_plan_set_timing_stride")

    def set_timing(self, slots):
        """Keep HIP-event timings of the last ``slots`` runs (0 switches recording off)."""
        self._eng._check(self._lib.ctcfa_plan_set_timing(self._h, int(slots)), "ctcfa_plan_set_timing")

    def get_timings(self, n):
        """-> (fill_ms[n], backtrack_ms[n]) of the last n runs (synchronises on them)."""
        a = (ctypes.c_float * n)()
        b = (ctypes.c_float * n)()
        self._eng._check(self._lib.ctcfa_plan_get_timings(self._h, int(n), a, b), "ctcfa_plan_get_timings")
        return np.array(a[:], np.float64), np.array(b[:], np.float64)

    def get_step_intervals(self, n):
        """-> ms[n - 1]: fill start of one recorded run to fill start of the next, over the last n recorded runs."""
        a = (ctypes.c_float * (n - 1))()
        self._eng._check(self._lib.ctcfa_plan_get_step_intervals(self._h, int(n), a), "ctcfa_plan_get_step_intervals")
        return np.array(a[:], np.float64)

    def run_device(self, d_lpz, d_labels, d_utt_begin, d_fol, d_char_prob, d_state, d_seg_start,
                   d_seg_end, d_seg_score, d_t_end, d_status, stream=None, pipelined=False):
        """All arguments are raw device addresses (ints) or None; ``stream`` a hipStream_t value."""
        args = [d_lpz, d_labels, d_utt_begin, d_fol, d_char_prob, d_state, d_seg_start, d_seg_end,
                d_seg_score, d_t_end, d_status, stream]
        fn = self._lib.ctcfa_plan_run_pipelined if pipelined else self._lib.ctcfa_plan_run_device
        rc = fn(self._h, *[ctypes.c_void_p(a) if a else None for a in args])
        self._eng._check(rc, "ctcfa_plan_run_pipelined" if pipelined else "ctcfa_plan_run_device")

    def bind(self, d_lpz, d_labels, d_utt_begin, d_fol, d_char_prob, d_state, d_seg_start, d_seg_end,
             d_seg_score, d_t_end, d_status, stream=None, pipelined=False):
        """``run_device`` with these arguments as a closure: the ctypes conversions are done once, not on
        every call (a dozen ``c_void_p`` objects cost as much as launching a small kernel)."""
        args = tuple(ctypes.c_void_p(a) if a else None for a in
                     (d_lpz, d_labels, d_utt_begin, d_fol, d_char_prob, d_state, d_seg_start, d_seg_end, d_seg_score,
                      d_t_end, d_status, stream))
        fn = self._lib.ctcfa_plan_run_pipelined if pipelined else self._lib.ctcfa_plan_run_device
        h, check, what = self._h, self._eng._check, "ctcfa_plan_run_pipelined" if pipelined else "ctcfa_plan_run_device"

        def run():
            rc = fn(h, *args)
            if rc:
                check(rc, what)
        return run

    def flush(self, stream=None):
        """Make ``stream`` wait for every backtrack a pipelined run left outstanding."""
        rc = self._lib.ctcfa_plan_flush(self._h, ctypes.c_void_p(stream) if stream else None)
        self._eng._check(rc, "ctcfa_plan_flush")
