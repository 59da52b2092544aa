#!/usr/bin/env python3
"""Generate tests/golden/anchor_traces.json by running the REFERENCE's own
get_file_iterative_segmentation (/root/reference/src/iterative_utterance_alignment.py:14-404)
and helper functions (/root/reference/src/utils/alignment_utils.py, text_utils.py) on
scripted inputs.  Runs only in the build container (the reference does not travel).

What is stubbed, and why this is still the reference's logic:
  * the reference's third-party imports that are absent here (torchaudio, speechbrain,
    num2words) are replaced by inert module objects so that the reference files IMPORT;
    none of their code is part of the logic under test;
  * audio I/O is scripted (torchaudio.info / torchaudio.load return fixed-length zeros);
  * the aligner is tests.fakes.ScriptedAligner (scores are a pure function of text and clip
    length) -- the same object drives this repo's loop in tests/test_anchor.py;
  * pandas >= 2 removed DataFrame.append, which alignment_utils.insert_row uses
    (alignment_utils.py:114): a 6-line compatibility shim restores the pandas-1.3 behaviour.
The fixture holds inputs and the reference's outputs only -- no reference source text.
"""
import json
import os
import sys
import types
import warnings

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(REF, "src"))
warnings.simplefilter("ignore")

from tests.fakes import ScriptedAligner, ScriptedASR  # noqa: E402

# ---- inert stand-ins for absent third-party modules ------------------------------------
_audio = {"seconds": 0.0, "sr": 16000, "fail_calls": (), "n_loads": 0}


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Info:
    def __init__(self):
        self.sample_rate = _audio["sr"]
        self.num_frames = int(_audio["seconds"] * _audio["sr"])


def _load(path, frame_offset=0, num_frames=-1, channels_first=False):
    import torch
    _audio["n_loads"] += 1
    if _audio["n_loads"] in _audio["fail_calls"]:
        raise RuntimeError("scripted load failure")
    total = int(_audio["seconds"] * _audio["sr"])
    n = max(0, min(num_frames if num_frames >= 0 else total, total - frame_offset))
    return torch.zeros(n, 1), _audio["sr"]


_mod("torchaudio", info=lambda p: _Info(), load=_load)
_mod("num2words", num2words=lambda n, lang="es": str(n))
_mod("speechbrain")
_mod("speechbrain.pretrained", EncoderASR=object)
_mod("speechbrain.alignment")
_mod("speechbrain.alignment.ctc_segmentation", CTCSegmentation=object)

if not hasattr(pd.DataFrame, "append"):  # pandas >= 2: restore the 1.3 semantics insert_row relies on
    def _append(self, other, ignore_index=False):
        if isinstance(other, pd.Series):
            other = other.to_frame().T
        return pd.concat([self, other], ignore_index=ignore_index)
    pd.DataFrame.append = _append

import iterative_utterance_alignment as ref_iua  # noqa: E402
from utils import alignment_utils as ref_au  # noqa: E402
from utils import text_utils as ref_tu  # noqa: E402

TSV = os.path.join(REF, "data", "sample", "tsv", "benedetti.tsv")


def scenario(name, n_rows, vad, audio_seconds, mode, salt=0, fail_loads=(), **kw):
    df = pd.read_csv(TSV, header=0, sep="\t").iloc[:n_rows].reset_index(drop=True)
    vad_df = pd.DataFrame([dict(Sample_ID=f"v{i}", Sample_Path=df.Sample_Path[0], Channel=1,
                                Audio_Length=e - s, Start=s, End=e, Segment_Length=e - s,
                                Transcription="Speech", Speaker_ID="u", Database="benedetti")
                           for i, (s, e) in enumerate(vad)])
    _audio["seconds"] = audio_seconds
    _audio["fail_calls"] = tuple(fail_loads)
    _audio["n_loads"] = 0
    os.makedirs("/tmp/anchor_golden_logs", exist_ok=True)
    aligner = ScriptedAligner(mode=mode, salt=salt)
    params = dict(threshold=-2.0, short_utterance_len=30, max_words_sequence=24, max_window_size=70.0,
                  window_to_stop=500.0, min_text_to_audio_prop=0.8, max_text_to_audio_prop_exec=10)
    params.update(kw)
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        out = ref_iua.get_file_iterative_segmentation(
            ScriptedASR(), aligner, df.Sample_Path[0], df.copy(), vad_df.copy(), 320.0,
            logs_path="/tmp/anchor_golden_logs", **params)
    rows = [[r[0], r[1], int(r[2]), float(r[3]), float(r[4]), float(r[5]), float(r[6]), r[7], r[8], r[9]] for r in out]
    return dict(name=name, n_rows=n_rows, vad=vad, audio_seconds=audio_seconds, mode=mode, salt=salt,
                fail_loads=list(fail_loads), params=params, file_alignments=rows, n_dp_calls=len(aligner.calls))


def helper_goldens():
    g = {}
    texts = ["uno dos tres", " ".join(f"w{i}" for i in range(60)), "a  b", "", " ".join(["x"] * 24), " ".join(["x"] * 25)]
    g["split_long_transcript"] = [[t, m, ref_tu.split_long_transcript(t, max_words_sequence=m)] for t in texts for m in (24, 5, 100)]
    g["prepare_text"] = [[t, m, ref_au.prepare_text(t, max_words_sequence=m)] for t in texts for m in (24, 5, 100)]
    g["count_text_length"] = [[t, ref_au.count_text_length(t)] for t in (["ab", "cde"], ["x"], [], ["hola que tal", "bien"])]
    g["get_text_to_audio_proportion"] = [[a, t, s, ref_au.get_text_to_audio_proportion(a, t, s)] for a, t, s in
                                         ((48000, 34, 16000), (160000, 10, 16000), (1, 1, 8000), (320000, 700, 16000))]
    g["get_n_aligned_rows"] = [[l, n, ref_au.get_n_aligned_rows(l, n)] for l, n in
                               (([1, 1, 1], 2), ([2, 1, 3, 1], 3), ([1], 0), ([3, 3], 7), ([1, 2, 1, 1, 1], 4))]
    import io, contextlib
    cases = []
    for audio_len, tr, ratio in ((16000, ["aaaa bbbb", "cc", "dddddd"], 320.0), (3200, ["aaaa bbbb", "cc"], 320.0),
                                 (320, ["abcdef"], 320.0), (64000, ["a" * 150, "b" * 60, "c" * 10], 320.0)):
        with contextlib.redirect_stdout(io.StringIO()):
            r = ref_au.find_a_valid_text_to_audio_proportion(audio_len, list(tr), ratio)
        cases.append([audio_len, tr, ratio, [list(r[0]), list(r[1])]])
    g["find_a_valid_text_to_audio_proportion"] = cases
    df = pd.DataFrame(dict(Transcription=["corto", "x" * 30, "y" * 29, "z" * 45], Segment_Score=[-5.0, -1.0, -6.5, -0.2]))
    g["remove_artefacts"] = [df.Transcription.tolist(), df.Segment_Score.tolist(), 30,
                             ref_au.remove_artefacts(df.copy(), 30).Segment_Score.tolist()]
    g["normalize_transcript"] = [[t, ref_tu.normalize_transcript(t)] for t in
                                 ('<font color="#00FF00">Hola, ¿qué tal?</font>', "¡Vamos!  a   ver...", "Línea\nnueva: fin")]
    # time references
    df = pd.read_csv(TSV, header=0, sep="\t").iloc[:12].reset_index(drop=True)
    tr = []
    for vad, real in (([(0.0, 60.0)], 61.5), ([(0.0, 10.0), (14.0, 30.0), (33.0, 50.0)], 55.0), ([(2.0, 20.0), (25.0, 40.0)], 38.0)):
        vad_df = pd.DataFrame([dict(Start=s, End=e, Segment_Length=e - s) for s, e in vad])
        out = ref_au.fix_time_reference(df.copy(), vad_df, real, len(df))
        tr.append(dict(vad=vad, real=real, rows=[[str(r.Sample_ID), float(r.Start), float(r.End), str(r.Type), int(r.Text_Length)]
                                                 for r in out.itertuples()]))
    g["fix_time_reference"] = tr
    return g


def main():
    scen = [
        scenario("good_single_vad", 10, [(0.0, 40.0)], 40.0, "good"),
        scenario("mixed_single_vad", 16, [(0.0, 60.0)], 61.0, "mixed", salt=1),
        scenario("mixed_two_vad", 20, [(0.0, 30.0), (36.0, 75.0)], 78.0, "mixed", salt=2),
        scenario("mixed_three_vad", 24, [(0.0, 25.0), (31.0, 60.0), (67.0, 95.0)], 97.0, "mixed", salt=3),
        scenario("bad_everything", 8, [(0.0, 30.0)], 30.0, "bad"),
        scenario("lastbad", 12, [(0.0, 45.0)], 46.0, "lastbad"),
        scenario("short_audio_exceptions", 14, [(0.0, 6.0)], 6.0, "mixed", salt=4),
        scenario("mixed_salt5_small_words", 18, [(0.0, 70.0)], 70.0, "mixed", salt=5, max_words_sequence=3),
        scenario("mixed_salt6_thr", 18, [(0.0, 50.0), (55.0, 80.0)], 82.0, "mixed", salt=6, threshold=-1.5, short_utterance_len=20),
        scenario("big_window", 30, [(0.0, 200.0)], 205.0, "bad", max_window_size=20.0),
        scenario("improving", 14, [(0.0, 60.0)], 60.0, "improving"),
        scenario("improving_small_chunks", 14, [(0.0, 60.0)], 60.0, "improving", max_words_sequence=4),
        scenario("improving_slow", 16, [(0.0, 70.0)], 71.0, "improving_slow"),
        scenario("improving_slow_chunks", 16, [(0.0, 70.0)], 71.0, "improving_slow", max_words_sequence=3, short_utterance_len=12),
        scenario("improving_thr", 16, [(0.0, 50.0)], 50.0, "improving", threshold=-1.5, max_words_sequence=5, short_utterance_len=18),
        scenario("stop_window", 20, [(0.0, 120.0)], 120.0, "bad", window_to_stop=25.0),
        scenario("vad_gaps_respread", 26, [(0.0, 12.0), (20.0, 34.0), (42.0, 58.0), (66.0, 90.0)], 92.0, "mixed", salt=7, max_window_size=9.0),
        scenario("tiny_speech_then_long", 22, [(0.0, 2.5), (12.0, 60.0)], 62.0, "mixed", salt=8),
        scenario("tiny_speech_good", 22, [(0.0, 2.0), (10.0, 15.0), (25.0, 80.0)], 82.0, "good"),
        scenario("improving_fast", 14, [(0.0, 60.0)], 60.0, "improving_fast"),
        scenario("improving_fast_chunks", 14, [(0.0, 60.0)], 60.0, "improving_fast", max_words_sequence=8, short_utterance_len=20),
        scenario("improving_fast_pairs", 14, [(0.0, 60.0)], 60.0, "improving_fast", max_words_sequence=4, short_utterance_len=8),
        scenario("improving_fast_pairs2", 16, [(0.0, 30.0), (38.0, 70.0)], 72.0, "improving_fast", max_words_sequence=5, short_utterance_len=6, threshold=-1.2),
        scenario("improving_fast_pairs3", 14, [(0.0, 60.0)], 60.0, "improving_fast", max_words_sequence=4, short_utterance_len=4),
        scenario("improving_fast_pairs4", 18, [(0.0, 40.0), (47.0, 80.0)], 82.0, "improving_fast", max_words_sequence=3, short_utterance_len=2, threshold=-1.6),
        scenario("dense_text_bad", 26, [(0.0, 6.0), (14.0, 20.0), (30.0, 34.0)], 36.0, "bad"),
        scenario("dense_text_mixed", 26, [(0.0, 6.0), (14.0, 20.0), (30.0, 34.0)], 36.0, "mixed", salt=11),
        scenario("dense_text_lastbad", 28, [(0.0, 7.0), (15.0, 22.0), (31.0, 37.0)], 40.0, "lastbad"),
        scenario("load_fails", 12, [(0.0, 45.0)], 46.0, "mixed", salt=9, fail_loads=(3, 7)),
    ] + [scenario(f"random_{k}", 10 + (k * 7) % 17, [(0.0, 20.0 + 3 * k), (24.0 + 3 * k, 70.0 + 2 * k)], 72.0 + 2 * k,
                  ("mixed", "improving_slow", "improving")[k % 3], salt=20 + k,
                  max_words_sequence=(24, 6, 3, 10)[k % 4], short_utterance_len=(30, 15, 10)[k % 3],
                  max_window_size=(70.0, 15.0, 30.0)[k % 3]) for k in range(12)]
    tsv_rows = pd.read_csv(TSV, header=0, sep="\t").iloc[:30].to_dict(orient="records")
    out = dict(tsv_rows=tsv_rows, provenance="outputs of /root/reference get_file_iterative_segmentation and utils, run with scripted aligner/audio",
               scenarios=scen, helpers=helper_goldens())
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "anchor_traces.json")
    json.dump(out, open(path, "w"), ensure_ascii=False, indent=0)
    print("wrote", path, os.path.getsize(path), "bytes;", {s["name"]: (len(s["file_alignments"]), s["n_dp_calls"]) for s in scen})


if __name__ == "__main__":
    main()
