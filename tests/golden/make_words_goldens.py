#!/usr/bin/env python3
"""Generate tests/golden/words_traces.json by running the REFERENCE's own
word_level_alignment.main (/root/reference/src/word_level_alignment.py:14-141) and
search_on_speech.main (/root/reference/src/search_on_speech.py:15-127) on scripted inputs.
Same approach and same stand-ins as make_anchor_goldens.py (absent third-party modules are
inert stubs, audio is scripted zeros, the aligner is tests.fakes.ScriptedAligner).  Build
container only.  The fixture holds the input rows and the rows of the TSVs the reference wrote.
"""
import contextlib
import io
import json
import os
import sys
import tempfile
import types

import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import tests.golden.make_anchor_goldens as base  # noqa: E402  (installs the stubs, imports reference utils)
from tests.fakes import ScriptedAligner, ScriptedASR  # noqa: E402

state = {"mode": "mixed", "salt": 0}


class _EncoderASR:
    @classmethod
    def from_hparams(cls, source=None, savedir=None, **kw):
        return ScriptedASR()


sys.modules["speechbrain.pretrained"].EncoderASR = _EncoderASR
sys.modules["speechbrain.alignment.ctc_segmentation"].CTCSegmentation = \
    lambda asr, **kw: ScriptedAligner(mode=state["mode"], salt=state["salt"])

import word_level_alignment as ref_wla  # noqa: E402
import search_on_speech as ref_sos  # noqa: E402


def input_rows(n, pick):
    df = pd.read_csv(base.TSV, header=0, sep="\t").iloc[:n].reset_index(drop=True)
    rows = []
    for i, r in enumerate(df.to_dict(orient="records")):
        norm = base.ref_tu.normalize_transcript(str(r["Transcription"])).upper()
        words = norm.split(" ")
        w = pick(i, words)
        rows.append(dict(r, Normalized_Transcription=norm, Wanted_Text=w))
    return rows


def run_words(name, rows, audio_seconds, mode, salt, **args):
    state.update(mode=mode, salt=salt)
    base._audio.update(seconds=audio_seconds, fail_calls=tuple(args.pop("fail_loads", ())), n_loads=0)
    tmp = tempfile.mkdtemp()
    tsv = os.path.join(tmp, "set_filtered.tsv")
    pd.DataFrame(rows).to_csv(tsv, sep="\t", index=None)
    ns = types.SimpleNamespace(tsv_path=tsv, dst_path=tmp, logs_path=tmp, asr_hub="", asr_savedir="",
                               time_info=True, offset_time=0.0, left_offset=0.0, right_offset=0.0, collar=0.0)
    ns.__dict__.update(args)
    with contextlib.redirect_stdout(io.StringIO()):
        ref_wla.main(ns)
    out = pd.read_csv(tsv.replace("_filtered.tsv", "_words.tsv"), header=0, sep="\t")
    return dict(name=name, rows=rows, audio_seconds=audio_seconds, mode=mode, salt=salt,
                args={k: v for k, v in ns.__dict__.items() if k in ("time_info", "offset_time", "left_offset", "right_offset")},
                fail_loads=list(base._audio["fail_calls"]), columns=list(out.columns), out=out.values.tolist())


def run_search(name, rows, audio_seconds, mode, salt, text, **args):
    state.update(mode=mode, salt=salt)
    base._audio.update(seconds=audio_seconds, fail_calls=(), n_loads=0)
    tmp = tempfile.mkdtemp()
    tsv = os.path.join(tmp, "clips.tsv")
    pd.DataFrame(rows).to_csv(tsv, sep="\t", index=None)
    ns = types.SimpleNamespace(tsv_path=tsv, dst_path=tmp, logs_path=tmp, asr_hub="", asr_savedir="", text=text,
                               offset_time=0.0, left_offset=0.0, right_offset=0.0, collar=0.0)
    ns.__dict__.update(args)
    with contextlib.redirect_stdout(io.StringIO()):
        ref_sos.main(ns)
    out = pd.read_csv(os.path.join(tmp, "clips_sos.tsv"), header=0, sep="\t")
    return dict(name=name, rows=rows, audio_seconds=audio_seconds, mode=mode, salt=salt, text=text,
                args={k: v for k, v in ns.__dict__.items() if k in ("offset_time", "left_offset", "right_offset")},
                columns=list(out.columns), out=out.values.tolist())


def main():
    second = lambda i, w: w[min(1, len(w) - 1)]
    last2 = lambda i, w: " ".join(w[-2:])
    first = lambda i, w: w[0]
    words = [
        run_words("second_word", input_rows(20, second), 600.0, "mixed", 1),
        run_words("last_two_words", input_rows(16, last2), 600.0, "mixed", 2, offset_time=0.1, left_offset=-0.05, right_offset=0.2),
        run_words("first_word_no_time_info", input_rows(12, first), 600.0, "good", 0, time_info=False),
        run_words("short_audio", input_rows(12, second), 14.0, "mixed", 3),
        run_words("load_fails_midway", input_rows(12, second), 600.0, "mixed", 4, fail_loads=(6,)),
    ]
    plain = pd.read_csv(base.TSV, header=0, sep="\t").iloc[:18].to_dict(orient="records")
    search = [
        run_search("mi_amor", plain, 600.0, "mixed", 5, "mi amor"),
        run_search("offsets", plain[:10], 600.0, "good", 0, "¡Horizonte!", offset_time=0.05, left_offset=-0.1, right_offset=0.1),
        run_search("short_audio", plain, 12.5, "mixed", 6, "queridas amigas quedas frágil en el horizonte he dejado pensando"),
    ]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "words_traces.json")
    json.dump(dict(provenance="outputs of /root/reference word_level_alignment.main / search_on_speech.main with scripted aligner/audio",
                   words=words, search=search), open(path, "w"), ensure_ascii=False, indent=0)
    print("wrote", path, os.path.getsize(path), "bytes", [(w["name"], len(w["out"])) for w in words],
          [(s["name"], len(s["out"])) for s in search])


if __name__ == "__main__":
    main()
