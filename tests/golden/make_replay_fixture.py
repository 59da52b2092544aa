#!/usr/bin/env python3
"""Generates tests/golden/benedetti_rows.json and tests/golden/replay_windows.json.

* benedetti_rows.json: the 157 rows (Start, End, Transcription, ids) of the reference's sample
  TSV /root/reference/data/sample/tsv/benedetti.tsv -- the "full long file" of BASELINE.json
  configs[1] (519 s).  Data, not source: the audio itself is not in the reference
  (data/sample/README.md:8-12), so the replay runs on seeded noise through a fake encoder.
* replay_windows.json: the sequence of DP calls (frames T, tokens per utterance) the anchor iteration
  issues for that file when every ``get_segments`` is answered by the CPU oracle -- the window
  sequence ``bench.py --workload replay`` times DP-only, and the draw distribution of
  ``--workload corpus`` (configs[4]).

Run here (needs /root/reference for the TSV; nothing of the reference is imported):
    python tests/golden/make_replay_fixture.py
"""
import csv
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import __graft_entry__ as ge  # noqa: E402
from oracle import oracle_c  # noqa: E402
from tests.fakes import FakeASR, oracle_backed  # noqa: E402
from tests.replay_common import REPLAY_AUDIO_SECONDS, REPLAY_PARAMS, NoiseAudio, replay_vad  # noqa: E402

SRC = "/root/reference/data/sample/tsv/benedetti.tsv"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    rows = []
    with open(SRC, newline="") as f:
        for r in csv.DictReader(f, delimiter="\t"):
            rows.append(dict(Sample_ID=r["Sample_ID"], Sample_Path=r["Sample_Path"], Channel=int(r["Channel"]),
                             Audio_Length=float(r["Audio_Length"]), Start=float(r["Start"]), End=float(r["End"]),
                             Transcription=r["Transcription"], Speaker_ID=r["Speaker_ID"], Database=r["Database"]))
    assert len(rows) == 157
    json.dump(dict(provenance="rows of /root/reference/data/sample/tsv/benedetti.tsv (data file of the reference)", rows=rows),
              open(os.path.join(HERE, "benedetti_rows.json"), "w"), ensure_ascii=False, indent=0)

    pkg = ge.build()
    anchor = importlib.import_module(pkg.__name__ + ".anchor")
    asr = FakeASR(seed=5, sharp=6.0)
    aligner = oracle_backed(pkg.CTCSegmentation(asr, kaldi_style_text=False, time_stamps="fixed", scoring_length=30), oracle_c)
    calls = []
    inner = aligner.get_segments_batch

    def recording(tasks, raise_errors=False):
        for t in tasks:
            ub = list(t.utt_begin_indices)
            calls.append(dict(T=int(t.lpz.shape[0]), utts=[int(ub[i + 1] - ub[i] - 1) for i in range(len(ub) - 1)],
                              C=int(len(t.ground_truth_mat))))
        return inner(tasks, raise_errors)
    aligner.get_segments_batch = recording
    params = anchor.AnchorParams(**REPLAY_PARAMS)
    co = anchor.file_alignment(asr, NoiseAudio(REPLAY_AUDIO_SECONDS, 2024), rows[0]["Sample_Path"], [dict(r) for r in rows],
                               replay_vad(), 320.0, params)
    out = anchor.run_batched([co], aligner)[0]
    T = np.array([c["T"] for c in calls])
    C = np.array([c["C"] for c in calls])
    print(f"{len(calls)} DP calls, {len(out)} result rows; T {T.min()}..{T.max()} (mean {T.mean():.0f}), "
          f"C {C.min()}..{C.max()} (mean {C.mean():.0f}); sum T = {T.sum()} frames")
    json.dump(dict(provenance="DP calls of anchor.file_alignment over benedetti_rows.json, fake encoder on seeded noise, "
                              "every get_segments answered by the CPU oracle (tests/golden/make_replay_fixture.py)",
                   audio_seconds=REPLAY_AUDIO_SECONDS, n_result_rows=len(out), calls=calls),
              open(os.path.join(HERE, "replay_windows.json"), "w"))


if __name__ == "__main__":
    main()
