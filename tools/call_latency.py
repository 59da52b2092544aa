#!/usr/bin/env python3
"""Tuning aid: wall time of ONE small alignment call through the Python protocol
(get_segments on a 10 s / 60 s window), i.e. what an anchor iteration pays per DP call."""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

pkg = ge.build()
DUR = 320.4769 / 16000
cfg = pkg.CtcSegmentationParameters(index_duration=DUR)
for T, U, n in ((499, 4, 25), (2999, 22, 28)):
    seg = pkg.synthetic.make_segment(3, T, 32, U, n)
    for _ in range(5):
        pkg.ctc_segmentation.get_segments_device(cfg, [seg[0]], [seg[1]], [seg[2]])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    N = 200
    for _ in range(N):
        pkg.ctc_segmentation.get_segments_device(cfg, [seg[0]], [seg[1]], [seg[2]])
    dt = (time.perf_counter() - t0) / N
    print(f"T={T} C={len(seg[1])}: {dt * 1e6:.0f} us per call (host arrays in, results out)", flush=True)
    # 16 windows in one call
    segs = [pkg.synthetic.make_segment(10 + i, T, 32, U, n) for i in range(16)]
    a, b, c = [s[0] for s in segs], [s[1] for s in segs], [s[2] for s in segs]
    for _ in range(3):
        pkg.ctc_segmentation.get_segments_device(cfg, a, b, c)
    t0 = time.perf_counter()
    for _ in range(50):
        pkg.ctc_segmentation.get_segments_device(cfg, a, b, c)
    dt = (time.perf_counter() - t0) / 50
    print(f"T={T} x16 windows: {dt * 1e6:.0f} us per call = {dt / 16 * 1e6:.0f} us per window", flush=True)
