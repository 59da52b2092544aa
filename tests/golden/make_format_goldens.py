#!/usr/bin/env python3
"""Generate tests/golden/format_traces.json by running the REFERENCE's own file-format steps
on small inputs: search_words.main (/root/reference/src/search_words.py:9-46),
tsv_to_stm.main (src/scripts/tsv_to_stm.py:6-42), merge_aligned_files.main
(src/postprocess/merge_aligned_files.py:7-30) and ptem.main (src/benchmark/ptem.py:10-64).
Same stand-ins as make_anchor_goldens.py (absent third-party modules are inert stubs, the
pandas >= 2 DataFrame.append shim).  Build container only.  The fixture holds input rows and
the text / rows of what the reference wrote or printed -- no reference source text.
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
import tests.golden.make_anchor_goldens as base  # noqa: E402  (stubs + reference utils on sys.path)

sys.path.insert(0, os.path.join(base.REF, "src", "scripts"))
sys.path.insert(0, os.path.join(base.REF, "src", "postprocess"))
sys.path.insert(0, os.path.join(base.REF, "src", "benchmark"))
import search_words as ref_sw  # noqa: E402
import tsv_to_stm as ref_stm  # noqa: E402
import merge_aligned_files as ref_merge  # noqa: E402
import ptem as ref_ptem  # noqa: E402
sys.path.insert(0, os.path.join(base.REF, "src", "preprocess"))
import filter_non_speech_segments as ref_vadf  # noqa: E402


def sample_rows(n):
    return pd.read_csv(base.TSV, header=0, sep="\t").iloc[:n].reset_index(drop=True)


def run_search_words(name, n_rows, words, text_column="Transcription"):
    tmp = tempfile.mkdtemp()
    df = sample_rows(n_rows)
    tsv = os.path.join(tmp, "part.tsv")
    df.to_csv(tsv, sep="\t", index=None)
    cfg = os.path.join(tmp, "words.json")
    json.dump({"words": words}, open(cfg, "w"))
    ns = types.SimpleNamespace(tsv_path=tsv, dst=tmp, config_file=cfg, text_column=text_column)
    with contextlib.redirect_stdout(io.StringIO()):
        ref_sw.main(ns)
    out = open(os.path.join(tmp, "part_filtered.tsv"), encoding="utf-8").read()
    return dict(name=name, rows=json.loads(df.to_json(orient="records")), columns=list(df.columns), words=words,
                text_column=text_column, out_tsv=out)


def run_stm(name, rows):
    src, dst = tempfile.mkdtemp(), tempfile.mkdtemp()
    pd.DataFrame(rows).to_csv(os.path.join(src, "prog_one.tsv"), sep="\t", index=None)
    ref_stm.main(types.SimpleNamespace(src_path=src, dst_path=dst))
    return dict(name=name, rows=rows, file="prog_one.tsv", out_stm=open(os.path.join(dst, "prog_one.stm"), encoding="utf-8").read())


def run_merge(name, global_rows, per_file):
    src = tempfile.mkdtemp()
    g = os.path.join(src, "train.tsv")
    pd.DataFrame(global_rows).to_csv(g, sep="\t", index=None)
    for fname, rows in per_file.items():
        pd.DataFrame(rows).to_csv(os.path.join(src, fname), sep="\t", index=None)
    ref_merge.main(types.SimpleNamespace(global_tsv=g, src=src))
    return dict(name=name, global_rows=global_rows, per_file=per_file,
                out_tsv=open(os.path.join(src, "train_aligned.tsv"), encoding="utf-8").read())


def run_ptem(name, ref_lines, hyp_lines, collar_ms):
    tmp = tempfile.mkdtemp()
    r, h = os.path.join(tmp, "ref.stm"), os.path.join(tmp, "hyp.stm")
    open(r, "w").write("".join(ref_lines))
    open(h, "w").write("".join(hyp_lines))
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        ref_ptem.main(["-r", r, "-h", h, "-c", str(collar_ms)])
    return dict(name=name, ref=ref_lines, hyp=hyp_lines, collar_ms=collar_ms, printed=buf.getvalue())


def run_vad_filter(name, rows, length):
    tmp = tempfile.mkdtemp()
    src = os.path.join(tmp, "vad.tsv")
    pd.DataFrame(rows).to_csv(src, sep="\t", index=None)
    ref_vadf.main(types.SimpleNamespace(src=src, dst=tmp, length=length))
    return dict(name=name, rows=rows, length=length, out_tsv=open(os.path.join(tmp, "vad_filtered.tsv"), encoding="utf-8").read())


def main():
    out = {"search_words": [], "stm": [], "merge": [], "ptem": [], "vad_filter": []}
    out["search_words"].append(run_search_words("two_words", 40, ["credo", "imágenes"]))
    out["search_words"].append(run_search_words("overlapping_words", 60, ["de", "que", "no"]))
    out["search_words"].append(run_search_words("no_match", 30, ["xyzzyx"] + ["uno"]))
    out["search_words"].append(run_search_words("all_words", 25, ["*"]))

    rows = [dict(Sample_ID=f"s{i}", Channel=1 + (i % 2), Speaker_ID=("unknown" if i % 3 else f"spk{i}"),
                 Start=round(1.23456 * i + 0.0004 * i * i, 5), End=round(1.23456 * i + 1.0 + 1e-4 * i, 5),
                 Transcription=t)
            for i, t in enumerate(["Credo", "DE PRONTO uno se aleja", "ñandú ÁRBOL", "a", "Hola, ¿qué tal?"])]
    out["stm"].append(run_stm("mixed_case_rounding", rows))

    samp = sample_rows(12)
    paths = ["data/a/one.wav", "data/a/two.wav", "data/b/three.wav"]
    grows = [dict(Sample_ID=f"g{i}", Sample_Path=paths[i % 3], Start=float(i), End=float(i) + 1.5, Transcription=str(samp["Transcription"][i]))
             for i in range(9)]
    per_file = {
        "one.tsv": [dict(Sample_ID=f"one{i}", Start=i * 1.0, End=i + 0.9, Score=-0.1 * i, Transcription=f"t{i}") for i in range(3)],
        "three.tsv": [dict(Sample_ID=f"three{i}", Start=i * 2.0, End=i * 2 + 1.1, Score=-0.25 * i, Transcription=f"u{i}") for i in range(2)],
    }
    out["merge"].append(run_merge("one_file_missing", grows, per_file))

    ref_l = [f"prog 1 spk {1.0 * i:.3f} {1.0 * i + 0.8:.3f} <,,> text {i}\n" for i in range(7)]
    hyp_l = [f"prog 1 spk {1.0 * i + 0.013 * (i % 4):.3f} {1.0 * i + 0.8 - 0.05 * (i % 3):.3f} <,,> text {i}\n" for i in range(7)]
    for collar in (0, 20, 60):
        out["ptem"].append(run_ptem(f"collar_{collar}", ref_l, hyp_l, collar))
    vad = []
    for f, (alen, segs) in {"data/x/a.wav": (300.0, [(2.0, 40.0), (41.5, 90.0), (130.0, 180.5), (181.0, 250.0), (290.0, 299.0)]),
                            "data/x/b.wav": (120.0, [(0.5, 30.0), (31.0, 60.0), (61.0, 119.0)]),
                            "data/y/c.wav": (500.0, [(100.0, 200.0), (260.0, 300.0), (400.0, 450.0)])}.items():
        for s0, e0 in segs:
            vad.append(dict(Sample_Path=f, Audio_Length=alen, Start=s0, End=e0, Segment_Length=round(e0 - s0, 3)))
    for length in (30.0, 5.0, 1000.0):
        out["vad_filter"].append(run_vad_filter(f"min_gap_{int(length)}", vad, length))
    path = os.path.join(ROOT, "tests", "golden", "format_traces.json")
    json.dump(out, open(path, "w"), ensure_ascii=False, indent=1)
    print("wrote", path, {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
