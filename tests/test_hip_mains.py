"""GPU: the PRODUCT mains (pipelines.utterance_main / word_main / search_main) with the acoustic model on
cuda:0 -- emissions stay in HBM (keep_lpz_on_device), rows go through in chunks, the
utterance stage runs its files in lockstep -- against the same mains answered by the oracle."""
import importlib
import os
import socket

import numpy as np
import pandas as pd
import pytest
import torch

from tests.fakes import FakeASR, oracle_backed
from tests.test_anchor import GOLD as ANCHOR_GOLD
from tests.test_hip_end_to_end import NoiseAudio

pytestmark = pytest.mark.gpu


def _host_emissions(aligner):
    """Test-only: the oracle wants NumPy emissions; the product aligner under test keeps them in HBM."""
    inner_one, inner_batch = aligner.get_lpz, aligner.get_lpz_batch
    aligner.get_lpz = lambda speech: inner_one(speech).cpu().numpy()
    aligner.get_lpz_batch = lambda speeches, frames_fn=None: [z.cpu().numpy() for z in inner_batch(speeches, frames_fn)]
    return aligner


def _aligners(pkg, oracle, asr, **kw):
    mk = lambda: pkg.CTCSegmentation(asr, kaldi_style_text=False, time_stamps="fixed", keep_lpz_on_device=True, **kw)
    return mk(), oracle_backed(_host_emissions(mk()), oracle)


def test_utterance_main_on_the_gpu_equals_the_oracle(pkg, oracle, tmp_path):
    pl = importlib.import_module(pkg.__name__ + ".pipelines")
    asr = FakeASR(seed=5, sharp=6.0, device="cuda:0")
    files = [("data/x/f%d.wav" % i, 12 + 2 * i, 60.0 + 7 * i, 40 + i) for i in range(5)]
    rows, vad = [], []
    for path, n, secs, _ in files:
        rows += [dict(r, Sample_Path=path) for r in ANCHOR_GOLD["tsv_rows"][:n]]
        vad += [dict(Sample_Path=path, Start=0.0, End=secs * 0.45, Segment_Length=secs * 0.45),
                dict(Sample_Path=path, Start=secs * 0.5, End=secs - 1.0, Segment_Length=secs * 0.5 - 1.0)]
    tsv, vtsv = str(tmp_path / "part.tsv"), str(tmp_path / "vad.tsv")
    pd.DataFrame(rows).to_csv(tsv, sep="\t", index=None)
    pd.DataFrame(vad).to_csv(vtsv, sep="\t", index=None)
    audio = {path: NoiseAudio(secs, seed) for path, _, secs, seed in files}
    hip, ref = _aligners(pkg, oracle, asr, scoring_length=30)
    assert hip.get_lpz(torch.zeros(16000)).is_cuda   # the product path keeps emissions in HBM

    def run(aligner, dst, extra=()):
        os.mkdir(dst)
        argv = ["--tsv", tsv, "--vad_segments_tsv", vtsv, "--dst", dst, "--threshold", "-6.0", "--short_utterance_len", "12",
                "--max_words_sequence", "6", "--files_per_round", "3", "--gather"] + list(extra)
        written = pl.utterance_main(pl.utterance_parser().parse_args(argv), asr, aligner, opener=lambda p: audio[p])
        assert len(written) == len(files)
        return pd.read_csv(os.path.join(dst, "part_aligned.tsv"), sep="\t")

    a = run(hip, str(tmp_path / "hip"))
    b = run(ref, str(tmp_path / "ref"))
    c = run(hip, str(tmp_path / "hip_batched"), ["--batch_lpz"])   # (the fake encoder is convolution-only: same frames)
    assert len(a) == len(b) > 30
    for got in (a, c):
        assert list(got.columns) == pl.UTT_COLUMNS and len(got) == len(b)
        for col in pl.UTT_COLUMNS:
            if col == "Segment_Score":
                np.testing.assert_allclose(got[col].values, b[col].values, rtol=0, atol=1e-4)
            else:
                assert got[col].tolist() == b[col].tolist(), col


def test_word_and_search_mains_on_the_gpu_equal_the_oracle(pkg, oracle, tmp_path):
    pl = importlib.import_module(pkg.__name__ + ".pipelines")
    tp = importlib.import_module(pkg.__name__ + ".text_prep")
    asr = FakeASR(seed=9, sharp=6.0, device="cuda:0")
    rows = []
    for i, r in enumerate(ANCHOR_GOLD["tsv_rows"][:30]):
        norm = tp.normalize_transcript(str(r["Transcription"])).upper()
        words = norm.split(" ")
        rows.append(dict(r, Normalized_Transcription=norm, Wanted_Text=words[min(1, len(words) - 1)],
                         Start=float(i), End=float(i) + 4.0 + (i % 3)))
    audio = NoiseAudio(45.0, 77)
    hip, ref = _aligners(pkg, oracle, asr)

    def words(aligner, name, extra=()):
        path = str(tmp_path / (name + "_filtered.tsv"))
        pd.DataFrame(rows).to_csv(path, sep="\t", index=None)
        args = pl.word_parser().parse_args(["--use_time_info", "--tsv_path", path, "--rows_per_launch", "7"] + list(extra))
        return pd.read_csv(pl.word_main(args, asr, aligner, opener=lambda p: audio), sep="\t")

    def search(aligner, name):
        path = str(tmp_path / (name + ".tsv"))
        pd.DataFrame(rows).to_csv(path, sep="\t", index=None)
        args = pl.word_parser(search=True).parse_args(["--tsv_path", path, "--dst_path", str(tmp_path), "--text", "horizonte",
                                                       "--rows_per_launch", "11"])
        return pd.read_csv(pl.search_main(args, asr, aligner, opener=lambda p: audio), sep="\t")

    for a, b, cols in ((words(hip, "w_hip"), words(ref, "w_ref"), pl.WORD_COLUMNS),
                       (words(hip, "w_hip_batched", ["--batch_lpz"]), words(ref, "w_ref2"), pl.WORD_COLUMNS),
                       (search(hip, "s_hip"), search(ref, "s_ref"), pl.SOS_COLUMNS)):
        assert list(a.columns) == cols and len(a) == len(b) >= 20
        for col in cols:
            if col == "Segment_Score":
                np.testing.assert_allclose(a[col].values, b[col].values, rtol=0, atol=1e-4)
            elif col != "Sample_Path":   # (the two runs read differently named input files)
                assert a[col].tolist() == b[col].tolist(), col


def test_gather_over_rccl_with_one_rank(pkg):
    """The collective of the row-level and utterance-level mains on its "nccl" (= RCCL) branch: a process
    group of one rank on cuda:0 (the multi-rank pattern runs on gloo in tests/test_sharding.py)."""
    import torch.distributed as dist
    pl = importlib.import_module(pkg.__name__ + ".pipelines")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        hits = [(5.0, 1.0, 0.25, 0.5, -1.5), (2.0, 0.0, 0.1, 0.9, -0.25), (9.0, 3.0, 1.5, 2.5, -3.0)]
        got = pl._gather_hits(dist, hits, 1)
        assert got == sorted(hits)
        sh = importlib.import_module(pkg.__name__ + ".sharding")
        part = sh.gather_records(torch.zeros(0, 12, dtype=torch.float64, device="cuda:0"), dist)
        assert len(part) == 1 and part[0].shape == (0, 12)
    finally:
        dist.destroy_process_group()
