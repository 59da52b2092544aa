#!/usr/bin/env python3
"""Tuning aid: config 3's batch (or another shape) under several settings of the environment knobs the plan reads when it
is created (CTCFA_TILE_PRIOS, CTCFA_PROD_PRIO, CTCFA_NS, CTCFA_SB_PRIO, CTCFA_SB_WAVES, ...) -- ONE process, the inputs built
once, a plan per setting: fill alone (serial schedule, HIP events), backtrack alone, and the pipelined step (wall clock
and the median of the fill-start intervals).

    python tools/env_sweep.py "base" "eq3:CTCFA_TILE_PRIOS=3" "t0hi:CTCFA_TILE_PRIOS=3,2,2,3,3,3" ...
    options: --segments N --frames T --vocab V --utts U --utt-len n --steps K --sets S --repeat R
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="*", default=["base"])
    ap.add_argument("--segments", type=int, default=512)
    ap.add_argument("--frames", type=int, default=3000)
    ap.add_argument("--vocab", type=int, default=32)
    ap.add_argument("--utts", type=int, default=22)
    ap.add_argument("--utt-len", type=int, default=28)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--sets", type=int, default=2)
    ap.add_argument("--repeat", type=int, default=1)
    ap.add_argument("--cols-per-lane", type=int, default=0)
    ap.add_argument("--alphabet", type=int, default=0, help="texts use the first N non-blank entries only")
    ap.add_argument("--narrow", action="store_true", help="plans are created with CTCFA_FLAG_TEXTS_OF_31_LABELS: vocabularies above 32 entries get a narrowed plan")
    ap.add_argument("--with-labels", action="store_true", help="plans are created with the labels of input set 0 (ctcfa_plan_create_shared): texts of up to 62 labels over more than 64 entries get the 64-entry ring")
    args = ap.parse_args()
    import torch
    import __graft_entry__ as ge
    pkg = ge.build()
    syn = pkg.synthetic
    B, T, V, U, n = args.segments, args.frames, args.vocab, args.utts, args.utt_len
    host = [syn.make_uniform_batch(B, T, V, U, n, seed0=64 * k * B, alphabet=args.alphabet or None) for k in range(args.sets)]
    C = host[0][1].shape[1]
    dev = torch.device("cuda:0")
    d_in = [(torch.from_numpy(h[0].reshape(-1)).to(dev), torch.from_numpy(h[1].astype(np.int32).reshape(-1)).to(dev),
             torch.from_numpy(h[2].astype(np.int32).reshape(-1)).to(dev)) for h in host]
    outs = [dict(fol=torch.empty(B * C, dtype=torch.int32, device=dev), cp=torch.empty(B * T, dtype=torch.float32, device=dev),
                 seg=torch.empty(3, B * U, dtype=torch.float64, device=dev), te=torch.empty(B, dtype=torch.int32, device=dev),
                 status=torch.empty(B, dtype=torch.int32, device=dev)) for _ in range(3)]
    stream = torch.cuda.current_stream().cuda_stream
    cfg = pkg.CtcSegmentationParameters(index_duration=0.02)
    eng = pkg._native.Engine(0)
    ref = None
    for rep in range(args.repeat):
        for spec in args.configs:
            name, _, envs = spec.partition(":")
            sets = dict(kv.split("=", 1) for kv in envs.split(";") if kv) if envs else {}
            old = {k: os.environ.get(k) for k in sets}
            os.environ.update(sets)
            try:
                plan = eng.plan(cfg.to_native(), V, [T] * B, [C] * B, [U] * B, force_cols_per_lane=args.cols_per_lane,
                                texts_of_31_labels=args.narrow,
                                labels=host[0][1].astype(np.int32).reshape(-1) if args.with_labels else None)
            finally:
                for k, v in old.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v

            def run(i, pipelined):
                a, b, c = d_in[i % len(d_in)]
                o = outs[i % 3]
                plan.run_device(a.data_ptr(), b.data_ptr(), c.data_ptr(), o["fol"].data_ptr(), o["cp"].data_ptr(), None,
                                o["seg"][0].data_ptr(), o["seg"][1].data_ptr(), o["seg"][2].data_ptr(), o["te"].data_ptr(),
                                o["status"].data_ptr(), stream, pipelined=pipelined)

            K = args.steps
            for i in range(200):   # clocks
                run(i, False)
            torch.cuda.synchronize()
            plan.set_timing(64)
            plan.set_timing_stride(1)
            for i in range(64):
                run(i, False)
            torch.cuda.synchronize()
            fill, bt = plan.get_timings(64)
            plan.set_timing(0)
            for i in range(50):
                run(i, True)
            plan.flush(stream)
            torch.cuda.synchronize()
            nt = min(K // 4, 256)
            plan.set_timing(max(nt, 8))
            plan.set_timing_stride(4)
            t0 = time.perf_counter()
            for i in range(K):
                run(i, True)
            plan.flush(stream)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / K * 1e3
            pf, pb = plan.get_timings(nt)
            iv = plan.get_step_intervals(nt) / 4
            # results of the last pipelined step: equal to the first configuration's (a wrong priority table must not change them)
            last = {k: v.cpu().numpy().copy() for k, v in outs[(K - 1) % 3].items()}
            ok = bool((last["status"] == 0).all())
            if ref is None:
                ref = last
            same = all(np.array_equal(ref[k], last[k]) for k in ("fol", "te", "seg"))
            print("%-14s fill alone %6.1f us (min %6.1f)  backtrack alone %6.1f us | pipelined: step %7.4f ms wall, %7.4f median of events; "
                  "fill %6.1f us, backtrack %6.1f us | status ok %s, == first config %s  [K=%d W=%d]"
                  % (name, np.mean(fill) * 1e3, np.min(fill) * 1e3, np.mean(bt) * 1e3, dt, float(np.median(iv)), np.mean(pf) * 1e3,
                     np.mean(pb) * 1e3, ok, same, plan.info["cols_per_lane"], plan.info["waves_per_seg"]), flush=True)
            plan.close()


if __name__ == "__main__":
    main()
