#!/usr/bin/env python3
"""Tuning aid: kernel times of one small window (T = 499) under forced trace modes / tile widths.
   tools/small_modes.py strider : checkpoint mode only, over the strider's wave and window counts (CTCFA_SB_WAVES / _WINDOWS)"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

pkg = ge.build()
cfg = pkg.CtcSegmentationParameters(index_duration=320.4769 / 16000)
cs = pkg.ctc_segmentation
STRIDER = len(sys.argv) > 1 and sys.argv[1] == "strider"
SHAPES = ((499, 4, 25), (318, 2, 20), (700, 6, 20))
if len(sys.argv) > 1 and sys.argv[1] == "short":   # where the lone-launch rule should stop
    SHAPES = ((96, 1, 12), (160, 1, 20), (224, 2, 15), (288, 2, 20), (352, 3, 18))
for T, U, n in SHAPES:
    seg = pkg.synthetic.make_segment(3, T, 32, U, n)
    for mode in ([f"ckpt waves={w} windows={x}" for w in (2, 3, 5, 7) for x in (1, 2, 3) if x <= w] if STRIDER else ("auto", "ckpt", "dec")):
        os.environ.pop("CTCFA_CHECKPOINT", None)
        os.environ.pop("CTCFA_DECISION_BITS", None)
        if mode.startswith("ckpt"):
            os.environ["CTCFA_CHECKPOINT"] = "1"
        if STRIDER:
            os.environ["CTCFA_SB_WAVES"] = mode.split("waves=")[1].split()[0]
            os.environ["CTCFA_SB_WINDOWS"] = mode.split("windows=")[1]
        if mode == "dec":
            os.environ["CTCFA_DECISION_BITS"] = "1"
        for K in ((0,) if STRIDER or len(sys.argv) > 1 else (0, 1, 2)):
            eng = cs.default_engine()
            try:
                plan = eng.plan(cfg.to_native(), 32, [T], [len(seg[1])], [len(seg[2]) - 1], force_cols_per_lane=K)
            except Exception as e:
                print(T, mode, K, "n/a", e)
                continue
            dev = torch.device("cuda:0")
            d_lpz = torch.from_numpy(seg[0].reshape(-1)).to(dev)
            d_lab = torch.from_numpy(seg[1].astype(np.int32)).to(dev)
            d_ub = torch.from_numpy(seg[2].astype(np.int32)).to(dev)
            fol = torch.empty(len(seg[1]), dtype=torch.int32, device=dev)
            cp = torch.empty(T, dtype=torch.float32, device=dev)
            sg = torch.empty(3, len(seg[2]) - 1, dtype=torch.float64, device=dev)
            te = torch.empty(1, dtype=torch.int32, device=dev)
            st = torch.empty(1, dtype=torch.int32, device=dev)
            stream = torch.cuda.current_stream().cuda_stream
            step = plan.bind(d_lpz.data_ptr(), d_lab.data_ptr(), d_ub.data_ptr(), fol.data_ptr(), cp.data_ptr(), None,
                             sg[0].data_ptr(), sg[1].data_ptr(), sg[2].data_ptr(), te.data_ptr(), st.data_ptr(), stream)
            for _ in range(200):
                step()
            torch.cuda.synchronize()
            plan.set_timing(50)
            for _ in range(50):
                step()
                torch.cuda.synchronize()
            f, b = plan.get_timings(50)
            info = plan.info
            print(f"T={T} C={len(seg[1])} {mode} K={info['cols_per_lane']} W={info['waves_per_seg']}: fill {np.median(f) * 1e3:.1f} us + "
                  f"backtrack {np.median(b) * 1e3:.1f} us = {(np.median(f) + np.median(b)) * 1e3:.1f}", flush=True)
            plan.close()
