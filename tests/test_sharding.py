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
sys.path.insert(0, {root!r})
import numpy as np, pandas as pd, torch
import torch.distributed as dist
import __graft_entry__ as ge
from tests.fakes import ScriptedAligner, ScriptedASR
from tests.test_anchor import ZeroAudio
pkg = ge.build()
pl = importlib.import_module(pkg.__name__ + ".pipelines")
sh = importlib.import_module(pkg.__name__ + ".sharding")
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
case = json.load(open({gold!r}))["words"][0]
df = pd.DataFrame(case["rows"])
costs = [(float(r["End"]) - float(r["Start"])) for r in case["rows"]]
units = sh.assign_units(costs, world)
mine = df.iloc[units[rank]].reset_index(drop=True)
rows = pl.align_words(ScriptedASR(), ScriptedAligner(mode=case["mode"], salt=case["salt"]), mine,
                      opener=lambda p: ZeroAudio(case["audio_seconds"]), **case["args"])
# fixed-width records: unit id, start, end, score  (text columns stay with the unit table)
recs = [[float(units[rank][i]), r[3], r[4], r[5]] for i, r in enumerate(rows)] if len(rows) == len(mine) else None
assert recs is not None, "every row of this fixture yields exactly one word segment"
local = torch.tensor(recs, dtype=torch.float64).reshape(-1, 4)
gathered = sh.gather_records(local, dist)
merged = sh.merge_in_unit_order(gathered, units)
if rank == 0:
    json.dump(dict(merged=merged, units=units), open({out!r}, "w"))
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


def test_two_rank_gloo_word_alignment_equals_single_process(pkg, tmp_path):
    gold = os.path.join(ROOT, "tests", "golden", "words_traces.json")
    out = str(tmp_path / "merged.json")
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, gold=gold, out=out))
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    res = json.load(open(out))
    case = json.load(open(gold))["words"][0]
    # single-process reference result (the reference's own output TSV rows, in row order)
    assert len(res["merged"]) == len(case["out"])
    for rec, ref in zip(res["merged"], case["out"]):
        assert rec[1] == ref[3] and rec[2] == ref[4] and rec[3] == ref[5]
    assert [int(r[0]) for r in res["merged"]] == list(range(len(case["out"])))
