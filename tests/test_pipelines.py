"""CPU suite: word-level and search-on-speech row logic + TSV I/O against the REFERENCE's
own outputs (tests/golden/words_traces.json from tests/golden/make_words_goldens.py), and the
utterance-level stage end to end with the scripted aligner."""
import importlib
import json
import os

import numpy as np
import pandas as pd
import pytest
import torch

from tests.fakes import ScriptedAligner, ScriptedASR
from tests.test_anchor import GOLD as ANCHOR_GOLD, ZeroAudio, _vad_rows

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "words_traces.json")))


def _pipelines(pkg):
    return importlib.import_module(pkg.__name__ + ".pipelines")


def _opener(seconds, fail_calls=()):
    shared = ZeroAudio(seconds, fail_calls=fail_calls)   # one counter across files, like the stubbed torchaudio.load
    return lambda path: shared


def _same(a, b):
    if isinstance(a, float) or isinstance(b, float):
        return float(a) == float(b) or (np.isnan(float(a)) and np.isnan(float(b)))
    return a == b


@pytest.mark.parametrize("case", GOLD["words"], ids=[c["name"] for c in GOLD["words"]])
def test_word_level_matches_reference(pkg, case, tmp_path):
    pl = _pipelines(pkg)
    df = pd.DataFrame(case["rows"])
    rows = pl.align_words(ScriptedASR(), ScriptedAligner(mode=case["mode"], salt=case["salt"]), df,
                          opener=_opener(case["audio_seconds"], case["fail_loads"]), **case["args"])
    out = tmp_path / "x_words.tsv"
    pl.write_tsv(str(out), rows, pl.WORD_COLUMNS)
    got = pd.read_csv(out, header=0, sep="\t")
    assert list(got.columns) == case["columns"]
    assert len(got) == len(case["out"])
    for a, b in zip(got.values.tolist(), case["out"]):
        assert all(_same(x, y) for x, y in zip(a, b)), (a, b)


@pytest.mark.parametrize("case", GOLD["search"], ids=[c["name"] for c in GOLD["search"]])
def test_search_on_speech_matches_reference(pkg, case, tmp_path):
    pl = _pipelines(pkg)
    df = pd.DataFrame(case["rows"])
    rows = pl.search_on_speech(ScriptedASR(), ScriptedAligner(mode=case["mode"], salt=case["salt"]), df, case["text"],
                               opener=_opener(case["audio_seconds"]), **case["args"])
    out = tmp_path / "x_sos.tsv"
    pl.write_tsv(str(out), rows, pl.SOS_COLUMNS)
    got = pd.read_csv(out, header=0, sep="\t")
    assert list(got.columns) == case["columns"] and len(got) == len(case["out"])
    for a, b in zip(got.values.tolist(), case["out"]):
        assert all(_same(x, y) for x, y in zip(a, b)), (a, b)


def test_sentence_pieces(pkg):
    pl = _pipelines(pkg)
    assert pl.sentence_pieces("HOLA MI AMOR QUE TAL", "MI AMOR") == ["HOLA", "·", "MI AMOR", "·", "QUE TAL", "·"]
    assert pl.sentence_pieces("MI AMOR QUE TAL", "MI AMOR") == ["MI AMOR", "·", "QUE TAL", "·"]
    assert pl.sentence_pieces("HOLA MI AMOR", "MI AMOR") == ["HOLA", "·", "MI AMOR", "·"]


def test_utterance_stage_end_to_end(pkg, tmp_path):
    """TSV in -> per-file result TSV out, resume rule, rank sharding, artefact score restore."""
    pl = _pipelines(pkg)
    anchor = importlib.import_module(pkg.__name__ + ".anchor")
    sc = [s for s in ANCHOR_GOLD["scenarios"] if s["name"] == "mixed_two_vad"][0]
    base = [dict(r) for r in ANCHOR_GOLD["tsv_rows"][: sc["n_rows"]]]
    files = ["data/a/file0.wav", "data/a/file1.wav", "data/a/file2.wav"]
    df = pd.DataFrame([dict(r, Sample_Path=f) for f in files for r in base])
    vad = pd.DataFrame([dict(Sample_Path=f, Start=s, End=e, Segment_Length=e - s) for f in files for s, e in sc["vad"]])
    dst = tmp_path / "results"
    dst.mkdir()
    params = anchor.AnchorParams(**sc["params"])
    mk = lambda: ScriptedAligner(mode=sc["mode"], salt=sc["salt"])
    opener = lambda path: ZeroAudio(sc["audio_seconds"])
    # two "ranks" shard the three files deterministically
    w0 = pl.align_utterance_files(ScriptedASR(), mk(), df, vad, str(dst), "", params, opener, rank=0, world=2)
    w1 = pl.align_utterance_files(ScriptedASR(), mk(), df, vad, str(dst), "", params, opener, rank=1, world=2)
    assert sorted(os.path.basename(p) for p in w0 + w1) == ["file0.tsv", "file1.tsv", "file2.tsv"]
    assert len(w0) == 2 and len(w1) == 1
    got = pd.read_csv(dst / "file1.tsv", header=0, sep="\t")
    assert list(got.columns) == pl.UTT_COLUMNS
    ref = sc["file_alignments"]
    assert len(got) == len(ref)
    for a, b in zip(got.values.tolist(), ref):
        score = b[6] + (4.0 if len(b[7]) < params.short_utterance_len else 0.0)   # remove_artefacts
        # (pandas' default CSV float parser is not round-trip exact: compare to 1e-9)
        assert a[4] == pytest.approx(b[4], abs=1e-9) and a[5] == pytest.approx(b[5], abs=1e-9) and a[7] == b[7]
        assert a[6] == pytest.approx(score, abs=1e-9)
    # resume: nothing is redone when the result files exist
    assert pl.align_utterance_files(ScriptedASR(), mk(), df, vad, str(dst), "", params, opener) == []


def test_wav_reader(pkg, tmp_path):
    import wave
    pl = _pipelines(pkg)
    path = tmp_path / "t.wav"
    data = (np.sin(np.arange(16000) / 10.0) * 20000).astype(np.int16)
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(data.tobytes())
    f = pl.WavFile(str(path))
    assert (f.num_frames, f.sample_rate) == (16000, 16000)
    clip, sr = f.load(4000, 8000)
    assert clip.shape == (8000, 1) and sr == 16000
    np.testing.assert_allclose(clip[:, 0].numpy(), data[4000:12000] / 32768.0, atol=1e-7)


def test_utterance_stage_failure_costs_one_file_and_resume_redoes_empty_results(pkg, tmp_path, capsys):
    """A file that cannot be aligned costs that file only (no empty result left behind, the others of
    its lockstep round are written); a 0-byte result TSV -- what a killed run of the reference leaves
    (align_utterances.sh:105-107) -- is not taken for a finished file."""
    pl = _pipelines(pkg)
    anchor = importlib.import_module(pkg.__name__ + ".anchor")
    sc = [s for s in ANCHOR_GOLD["scenarios"] if s["name"] == "mixed_two_vad"][0]
    base = [dict(r) for r in ANCHOR_GOLD["tsv_rows"][: sc["n_rows"]]]
    files = ["data/a/good0.wav", "data/a/broken.wav", "data/a/good1.wav"]
    df = pd.DataFrame([dict(r, Sample_Path=f) for f in files for r in base])
    vad = pd.DataFrame([dict(Sample_Path=f, Start=s, End=e, Segment_Length=e - s) for f in files for s, e in sc["vad"]])
    dst = tmp_path / "results"
    dst.mkdir()
    (dst / "good1.tsv").write_text("")   # stale empty result of an earlier, killed run
    params = anchor.AnchorParams(**sc["params"])

    class Broken(ZeroAudio):
        num_frames = property(lambda self: (_ for _ in ()).throw(RuntimeError("unreadable audio")), lambda self, v: None)

    def opener(path):
        if "broken" in path:
            return Broken(sc["audio_seconds"])
        return ZeroAudio(sc["audio_seconds"])
    written = pl.align_utterance_files(ScriptedASR(), ScriptedAligner(mode=sc["mode"], salt=sc["salt"]), df, vad,
                                       str(dst), "", params, opener)
    assert sorted(os.path.basename(p) for p in written) == ["good0.tsv", "good1.tsv"]
    assert not (dst / "broken.tsv").exists()
    assert (dst / "good1.tsv").stat().st_size > 0
    assert not [f for f in os.listdir(dst) if ".tmp." in f]
    assert "broken.wav" in capsys.readouterr().out
    # resume: the two finished files are skipped, the broken one is tried again (and fails again)
    again = pl.align_utterance_files(ScriptedASR(), ScriptedAligner(mode=sc["mode"], salt=sc["salt"]), df, vad,
                                     str(dst), "", params, opener)
    assert again == []


def test_a_sharded_word_run_stops_where_the_reference_stops(pkg):
    """The reference's single process ends the word-level run at the first unreadable clip
    (word_level_alignment.py:63-66).  Sharded, every rank reports the row that stopped IT and rank 0 keeps only the
    hits before the smallest such row -- the table a single process would have written, whatever the world size."""
    pl = _pipelines(pkg)
    case = GOLD["words"][0]
    records = pd.DataFrame(case["rows"]).to_dict(orient="records")
    n = len(records)
    assert n >= 6
    bad = n // 2

    class Picky(ZeroAudio):
        def __init__(self, seconds, fails):
            super().__init__(seconds)
            self._fails = fails

        def load(self, frame_offset, num_frames):
            if self._fails:
                raise RuntimeError("unreadable")
            return super().load(frame_offset, num_frames)
    for r in records:
        r["Sample_Path"] = "x/%d.wav" % records.index(r)
    opener = lambda path: Picky(case["audio_seconds"], path == "x/%d.wav" % bad)
    mk = lambda: ScriptedAligner(mode=case["mode"], salt=case["salt"])
    single, stop = pl.word_hits(ScriptedASR(), mk(), records, range(n), opener, return_stop=True, **case["args"])
    assert stop == bad and all(h[0] < bad for h in single)
    # two "ranks": even rows / odd rows; the rank that does not own the bad row runs to the end of its share
    shares = [list(range(0, n, 2)), list(range(1, n, 2))]
    parts = [pl.word_hits(ScriptedASR(), mk(), records, sh, opener, return_stop=True, **case["args"]) for sh in shares]
    stops = [s for _, s in parts if s is not None]
    assert stops == [bad]
    merged = pl._gather_hits(None, [h for hits, _ in parts for h in hits], 1, min(stops))
    assert merged == sorted(single)
    with pytest.raises(RuntimeError):
        pl._gather_hits(None, [], 1, None, RuntimeError("rank failed"))
