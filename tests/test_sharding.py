"""CPU suite: the N>1 path with world_size-2 gloo processes: deterministic unit
assignment, sharded word-level alignment, gather of result records, merge == 1-process run."""
import importlib
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import importlib, json, os, sys
from types import SimpleNamespace
sys.path.insert(0, {root!r})
import pandas as pd
import __graft_entry__ as ge
from tests.fakes import ScriptedAligner, ScriptedASR
from tests.test_anchor import ZeroAudio
pkg = ge.build()
pl = importlib.import_module(pkg.__name__ + ".pipelines")
gold = json.load(open({gold!r}))
# the PRODUCT mains, one process per rank (RANK / WORLD_SIZE / MASTER_* from the environment; gloo on CPU)
case = gold["words"][{wcase}]
args = SimpleNamespace(tsv_path={wtsv!r}, logs_path="", **case["args"])
w = pl.word_main(args, ScriptedASR(), ScriptedAligner(mode=case["mode"], salt=case["salt"]),
                 opener=lambda p: ZeroAudio(case["audio_seconds"]))
case = gold["search"][{scase}]
args = SimpleNamespace(tsv_path={stsv!r}, dst_path={dst!r}, logs_path="", text=case["text"], **case["args"])
s = pl.search_main(args, ScriptedASR(), ScriptedAligner(mode=case["mode"], salt=case["salt"]),
                   opener=lambda p: ZeroAudio(case["audio_seconds"]))
rank = int(os.environ["RANK"])
assert (w is None) == (rank != 0) and (s is None) == (rank != 0), "only rank 0 writes"
import torch.distributed as dist
dist.barrier()
dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_assign_units_is_deterministic_and_balanced(pkg):
    sh = importlib.import_module(pkg.__name__ + ".sharding")
    costs = [5, 1, 9, 3, 3, 7, 2, 8]
    a = sh.assign_units(costs, 3)
    assert a == sh.assign_units(list(costs), 3)
    assert sorted(i for b in a for i in b) == list(range(8))
    loads = [sum(costs[i] for i in b) for b in a]
    assert max(loads) - min(loads) <= max(costs)
    assert sh.assign_units(costs, 1) == [list(range(8))]


def _same(a, b):
    if isinstance(a, float) or isinstance(b, float):
        return float(a) == float(b) or (np.isnan(float(a)) and np.isnan(float(b)))
    return a == b


@pytest.mark.parametrize("wcase,scase", [(0, 0), (1, 1), (3, 2)])
def test_two_rank_gloo_product_mains_equal_the_reference(pkg, tmp_path, wcase, scase):
    """pipelines.word_main / search_main on two gloo ranks: rows sharded by cost, records gathered,
    rank 0 writes the TSVs -- byte-compatible with the reference's own single-process output."""
    import pandas as pd
    gold_path = os.path.join(ROOT, "tests", "golden", "words_traces.json")
    gold = json.load(open(gold_path))
    wtsv = str(tmp_path / "corpus_filtered.tsv")
    stsv = str(tmp_path / "segments.tsv")
    pd.DataFrame(gold["words"][wcase]["rows"]).to_csv(wtsv, sep="\t", index=None)
    pd.DataFrame(gold["search"][scase]["rows"]).to_csv(stsv, sep="\t", index=None)
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, gold=gold_path, wcase=wcase, scase=scase, wtsv=wtsv, stsv=stsv,
                                    dst=str(tmp_path)))
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    for out, case in ((str(tmp_path / "corpus_words.tsv"), gold["words"][wcase]),
                      (str(tmp_path / "segments_sos.tsv"), gold["search"][scase])):
        got = pd.read_csv(out, header=0, sep="\t")
        assert list(got.columns) == case["columns"] and len(got) == len(case["out"])
        for a, b in zip(got.values.tolist(), case["out"]):
            assert all(_same(x, y) for x, y in zip(a, b)), (a, b)


def test_row_shards_cover_every_row_once(pkg):
    pl = importlib.import_module(pkg.__name__ + ".pipelines")
    sh = importlib.import_module(pkg.__name__ + ".sharding")
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "words_traces.json")))
    records = gold["words"][0]["rows"]
    costs = pl._row_costs(records)
    for world in (1, 2, 3, 8):
        units = sh.assign_units(costs, world)
        assert sorted(i for u in units for i in u) == list(range(len(records)))
        loads = [sum(costs[i] for i in u) for u in units]
        assert max(loads) - min(loads) <= max(costs)


UTT_WORKER = r'''
import importlib, json, os, sys
sys.path.insert(0, {root!r})
import pandas as pd
import __graft_entry__ as ge
from tests.fakes import ScriptedAligner, ScriptedASR
from tests.test_anchor import GOLD, ZeroAudio
pkg = ge.build()
pl = importlib.import_module(pkg.__name__ + ".pipelines")
sc = [s for s in GOLD["scenarios"] if s["name"] == "mixed_two_vad"][0]
argv = ["--tsv", {tsv!r}, "--vad_segments_tsv", {vad!r}, "--dst", {dst!r}, "--gather", "--files_per_round", "2"]
for k, v in sc["params"].items():
    if v is not None:
        argv += ["--" + k, str(v)]
args = pl.utterance_parser().parse_args(argv)
written = pl.utterance_main(args, ScriptedASR(), ScriptedAligner(mode=sc["mode"], salt=sc["salt"]),
                            opener=lambda p: ZeroAudio(sc["audio_seconds"]))
print("rank", os.environ["RANK"], "wrote", sorted(os.path.basename(w) for w in written))
import torch.distributed as dist
dist.barrier()
dist.destroy_process_group()
'''


def test_two_rank_gloo_utterance_main_gathers_the_merged_partition(pkg, tmp_path):
    """pipelines.utterance_main --gather on two gloo ranks: files sharded by cost, every rank writes the
    per-file TSVs of its files, the result records are all-gathered and rank 0 writes <name>_aligned.tsv --
    byte for byte what the reference's merge step (src/postprocess/merge_aligned_files.py:17-25, restated in
    formats.merge_aligned_files) makes of the per-file TSVs."""
    import filecmp
    import shutil

    import pandas as pd

    from tests.test_anchor import GOLD as ANCHOR_GOLD
    sc = [s for s in ANCHOR_GOLD["scenarios"] if s["name"] == "mixed_two_vad"][0]
    base = [dict(r) for r in ANCHOR_GOLD["tsv_rows"][: sc["n_rows"]]]
    files = ["data/a/file%d.wav" % i for i in range(5)]
    df = pd.DataFrame([dict(r, Sample_Path=f) for f in files for r in base])
    vad = pd.DataFrame([dict(Sample_Path=f, Start=s, End=e, Segment_Length=e - s) for f in files for s, e in sc["vad"]])
    dst = tmp_path / "results"
    dst.mkdir()
    tsv, vtsv = str(tmp_path / "train.tsv"), str(tmp_path / "vad.tsv")
    df.to_csv(tsv, sep="\t", index=None)
    vad.to_csv(vtsv, sep="\t", index=None)
    script = tmp_path / "utt_worker.py"
    script.write_text(UTT_WORKER.format(root=ROOT, tsv=tsv, vad=vtsv, dst=str(dst)))
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    assert "rank 0 wrote" in logs[0] and "rank 1 wrote" in logs[1]
    merged = dst / "train_aligned.tsv"
    assert merged.is_file()
    assert sorted(f for f in os.listdir(dst) if f.endswith(".tsv") and "aligned" not in f) == ["file%d.tsv" % i for i in range(5)]
    mine = tmp_path / "gathered.tsv"
    shutil.move(str(merged), str(mine))
    fm = importlib.import_module(pkg.__name__ + ".formats")
    ref = fm.merge_aligned_files(tsv, str(dst))
    assert filecmp.cmp(str(mine), ref, shallow=False)
    assert len(pd.read_csv(ref, sep="\t")) == 5 * len(sc["file_alignments"])
