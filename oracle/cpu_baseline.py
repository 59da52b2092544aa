#!/usr/bin/env python3
"""ORACLE -- TEST / MEASUREMENT INFRASTRUCTURE ONLY.  PARITY UNPINNED.

The ``cpu_baseline`` leg of bench.py as a process of its own: it is started BEFORE bench.py touches
the GPU (a process that has initialised HIP must not fork workers), times the C oracle on a bounded
sample of the SAME synthetic workload on

* one core                                   (BASELINE.md Baseline C, ``kind: "port"``),
* all the cores this process may use         (Baseline B: one worker process per core, disjoint
                                              shards -- the reference's ``n_process`` workers,
                                              /root/reference/align_utterances.sh:57,127-137),
* one core, "reference-structured"           (Baseline A: compiled fill + the NumPy twin's interpreted
                                              backtrack and scoring, a handful of segments),

and prints one JSON object.  Nothing here is on the product path.

    python -m oracle.cpu_baseline --workload synthetic|replay|words|corpus [--budget-s 10] ...
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
INDEX_DURATION = 320.4769 / 16000


def _synthetic_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "ctcfa_synthetic", os.path.join(ROOT, "iterative-pseudo-forced-alignment-ctc_amd", "synthetic.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def sample_segments(args):
    """A bounded sample of the workload bench.py runs on the GPU (same generators, same seeds)."""
    syn = _synthetic_module()
    if args.workload == "synthetic":
        n = min(args.sample, args.segments)
        return [syn.make_segment(b, args.frames, args.vocab, args.utts, args.utt_len) for b in range(n)], \
            f"{n} of the {args.segments} benchmark segments ({args.frames} frames, vocab {args.vocab})"
    calls = json.load(open(os.path.join(ROOT, "tests", "golden", "replay_windows.json")))["calls"]
    if args.workload == "replay":
        return syn.make_windows_like(calls, args.vocab), f"the {len(calls)} recorded DP calls of one file replay"
    if args.workload == "words":
        n = min(args.sample * 8, 4000)
        return syn.make_word_rows(n, args.vocab), f"{n} of the 10 000 word rows"
    ok = [c for c in calls if c["C"] <= c["T"]]
    drawn = syn.draw_corpus_calls(ok, int(args.sample * 3000))
    return syn.make_windows_like(drawn, args.vocab), f"{len(drawn)} windows drawn like the corpus stream"


def _worker(job):
    from oracle import oracle_c
    segs, budget = job
    cfg = oracle_c.make_config(index_duration=INDEX_DURATION)
    frames = sum(s[0].shape[0] for s in segs)
    t0, passes = time.perf_counter(), 0
    while True:
        oracle_c.time_ragged_batch(segs, cfg)
        passes += 1
        if time.perf_counter() - t0 >= budget or passes >= 4096:
            break
    return passes * frames, time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="synthetic")
    ap.add_argument("--segments", type=int, default=512)
    ap.add_argument("--frames", type=int, default=3000)
    ap.add_argument("--vocab", type=int, default=32)
    ap.add_argument("--utts", type=int, default=22)
    ap.add_argument("--utt-len", type=int, default=28)
    ap.add_argument("--sample", type=int, default=512, help="segments in the sample (synthetic); scales the others")
    ap.add_argument("--budget-s", type=float, default=8.0, help="CPU seconds per leg")
    ap.add_argument("--max-workers", type=int, default=64)
    args = ap.parse_args()
    from oracle import oracle_c
    oracle_c.build()
    segs, what = sample_segments(args)
    frames = sum(s[0].shape[0] for s in segs)
    cfg = oracle_c.make_config(index_duration=INDEX_DURATION)

    # ---- one core (compiled fill + backtrack + scoring: the C port) ----
    sec, reps = 0.0, 0
    while sec < args.budget_s and reps < 4096:
        dt, _ = oracle_c.time_ragged_batch(segs, cfg)
        sec += dt
        reps += 1
    fps1 = reps * frames / sec
    out = {"value": fps1 * INDEX_DURATION / 3600.0, "unit": "audio-hours/s", "frames_per_s": fps1, "cores": 1, "kind": "port",
           "sample": f"{reps} pass(es) over {what}, oracle/ctc_segmentation_oracle.c, single thread, {sec:.1f} s"}

    # ---- all cores: one worker process per core this process may run on, disjoint shards ----
    try:
        ncpu = len(os.sched_getaffinity(0))
    except Exception:
        ncpu = os.cpu_count() or 1
    workers = max(1, min(ncpu, args.max_workers, len(segs)))
    shards = [segs[i::workers] for i in range(workers)]
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(workers) as pool:
        done = pool.map(_worker, [(sh, args.budget_s * 0.6) for sh in shards])
    wall = time.perf_counter() - t0
    fps_all = sum(f / t for f, t in done)          # every worker's own rate (they run concurrently)
    out["all_cores"] = {"value": fps_all * INDEX_DURATION / 3600.0, "unit": "audio-hours/s", "frames_per_s": fps_all,
                        "cores": workers, "host_cores": os.cpu_count(), "kind": "port",
                        "sample": f"{workers} worker processes (one per usable core), disjoint shards of the same sample, "
                                  f"{wall:.1f} s wall"}

    # ---- "reference-structured": compiled fill + the package's interpreted backtrack / scoring ----
    try:
        from oracle import ctc_segmentation_twin as tw

        def _compiled_fill(table, lpz_b, gt_b, offsets, blank, flags):
            tb, offs, t_end = oracle_c.fill_table(lpz_b, gt_b, table.shape[0], blank, flags)
            table[...] = tb
            offsets[...] = offs
            return t_end, table.shape[1] - 1

        tw_fill, tw.cython_fill_table = tw.cython_fill_table, _compiled_fill
        conf = tw.CtcSegmentationParameters(index_duration=INDEX_DURATION)
        k, fr, t0 = 0, 0, time.perf_counter()
        while k < min(128, len(segs)) and time.perf_counter() - t0 < 6.0:   # (128 segments of the benchmark take ~1.5 s)
            lpz, gt, ub = segs[k]
            if len(gt) <= lpz.shape[0]:
                tim, cps, _ = tw.ctc_segmentation(conf, lpz, np.asarray(gt).reshape(-1, 1))
                tw.determine_utterance_segments(conf, ub, cps, tim, [""] * (len(ub) - 1))
                fr += lpz.shape[0]
            k += 1
        sec2 = time.perf_counter() - t0
        tw.cython_fill_table = tw_fill
        out["reference_structured"] = {"frames_per_s": fr / sec2, "value": fr / sec2 * INDEX_DURATION / 3600.0,
                                       "unit": "audio-hours/s", "cores": 1,
                                       "sample": f"{k} segments: compiled fill (C oracle) + interpreted backtrack and scoring "
                                                 f"(oracle/ctc_segmentation_twin.py), {sec2:.1f} s"}
    except Exception as exc:  # the headline baseline above does not depend on this
        out["reference_structured"] = {"error": repr(exc)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
