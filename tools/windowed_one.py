#!/usr/bin/env python3
"""Tuning aid: one windowed segment (T frames, U utterances of n labels) through the host-buffer entry, a few times --
for rocprofv3 --kernel-trace --stats (band_fill_kernel against windowed_kernel).  usage: windowed_one.py [T U n [reps]]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

pkg = ge.build()
a = [int(x) for x in sys.argv[1:]]
T, U, n = (a + [9500, 40, 30])[:3] if len(a) >= 3 else (9500, 40, 30)
reps = a[3] if len(a) > 3 else 5
seg = pkg.synthetic.make_segment(7, T, 32, U, n)
cfg = pkg.CtcSegmentationParameters(index_duration=0.02)
for i in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = pkg.ctc_segmentation.get_segments_device(cfg, [seg[0]], [seg[1]], [seg[2]])
    torch.cuda.synchronize()
    print("T=%d C=%d call %d: %.1f ms, status %d" % (T, len(seg[1]), i, (time.perf_counter() - t0) * 1e3, res[0]["status"]), flush=True)
