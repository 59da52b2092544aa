"""File formats / row filters either side of the path against goldens
produced by the reference's own scripts (tests/golden/make_format_goldens.py)."""
import io
import json
import os

import pandas as pd
import pytest

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "format_traces.json"), encoding="utf-8"))


def _num2words_stub(n, lang="es"):
    return str(n)   # the stand-in the golden generator used for the absent num2words package


@pytest.mark.parametrize("case", GOLD["search_words"], ids=lambda c: c["name"])
def test_search_words_filter(pkg, tmp_path, case):
    df = pd.DataFrame(case["rows"], columns=case["columns"])
    norm = lambda t: pkg.text_prep.normalize_transcript(t, number_to_words=_num2words_stub)
    filtered, counts = pkg.formats.filter_wanted_words(df, case["words"], case["text_column"], normalize=norm)
    buf = io.StringIO()
    filtered.to_csv(buf, sep="\t", index=None)
    assert buf.getvalue() == case["out_tsv"]
    assert int(counts.sum()) >= len(filtered.index)


def test_search_words_file_entry(pkg, tmp_path):
    case = GOLD["search_words"][0]
    tsv = tmp_path / "part.tsv"
    pd.DataFrame(case["rows"], columns=case["columns"]).to_csv(tsv, sep="\t", index=None)
    cfg = tmp_path / "words.json"
    cfg.write_text(json.dumps({"words": case["words"]}))
    out = pkg.formats.search_words(str(tsv), str(tmp_path), str(cfg),
                                   normalize=lambda t: pkg.text_prep.normalize_transcript(t, number_to_words=_num2words_stub))
    assert out.endswith("part_filtered.tsv") and open(out, encoding="utf-8").read() == case["out_tsv"]


@pytest.mark.parametrize("case", GOLD["stm"], ids=lambda c: c["name"])
def test_tsv_to_stm(pkg, tmp_path, case):
    src, dst = tmp_path / "src", tmp_path / "dst"
    src.mkdir()
    dst.mkdir()
    pd.DataFrame(case["rows"]).to_csv(src / case["file"], sep="\t", index=None)
    (src / ".hidden.tsv").write_text("x\n")
    written = pkg.formats.tsv_to_stm(str(src), str(dst))
    assert [os.path.basename(w) for w in written] == [case["file"].replace(".tsv", ".stm")]
    assert open(written[0], encoding="utf-8").read() == case["out_stm"]
    assert pkg.formats.tsv_to_stm(str(src), str(tmp_path / "missing")) == []


@pytest.mark.parametrize("case", GOLD["merge"], ids=lambda c: c["name"])
def test_merge_aligned_files(pkg, tmp_path, case):
    g = tmp_path / "train.tsv"
    pd.DataFrame(case["global_rows"]).to_csv(g, sep="\t", index=None)
    for fname, rows in case["per_file"].items():
        pd.DataFrame(rows).to_csv(tmp_path / fname, sep="\t", index=None)
    out = pkg.formats.merge_aligned_files(str(g), str(tmp_path))
    assert open(out, encoding="utf-8").read() == case["out_tsv"]
    assert pkg.formats.merge_aligned_files(str(tmp_path / "nope.tsv"), str(tmp_path)) is None


@pytest.mark.parametrize("case", GOLD["ptem"], ids=lambda c: c["name"])
def test_ptem(pkg, tmp_path, case):
    r, h = tmp_path / "ref.stm", tmp_path / "hyp.stm"
    r.write_text("".join(case["ref"]))
    h.write_text("".join(case["hyp"]))
    assert pkg.formats.ptem_report(str(r), str(h), case["collar_ms"]) == case["printed"]
    m = pkg.formats.ptem(str(r), str(h), case["collar_ms"])
    assert m["n_ref"] == m["n_hyp"] == len(case["ref"])


def test_stm_round_trip_feeds_ptem(pkg, tmp_path):
    """aligned TSV -> STM -> PTEM against itself is exactly zero."""
    rows = GOLD["stm"][0]["rows"]
    src, dst = tmp_path / "s", tmp_path / "d"
    src.mkdir()
    dst.mkdir()
    pd.DataFrame(rows).to_csv(src / "a.tsv", sep="\t", index=None)
    stm = pkg.formats.tsv_to_stm(str(src), str(dst))[0]
    m = pkg.formats.ptem(stm, stm, 0)
    assert m["mean_program"] == 0.0 and m["median_program"] == 0.0


@pytest.mark.parametrize("case", GOLD["vad_filter"], ids=lambda c: c["name"])
def test_filter_non_speech_segments(pkg, tmp_path, case):
    src = tmp_path / "vad.tsv"
    pd.DataFrame(case["rows"]).to_csv(src, sep="\t", index=None)
    out = pkg.formats.filter_vad_file(str(src), str(tmp_path), case["length"])
    assert open(out, encoding="utf-8").read() == case["out_tsv"]
