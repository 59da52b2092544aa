#!/usr/bin/env python3
"""Tuning aid: wall time of the windowed regime (T > min_window_size = 8000) on the GPU vs the
C oracle on one host core, for a few window lengths."""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import __graft_entry__ as ge  # noqa: E402
from oracle import oracle_c as oc  # noqa: E402

pkg = ge.build()
DUR = 320.4769 / 16000
for T, U, n in ((9500, 40, 30), (16000, 100, 29), (25000, 200, 29)):
    seg = pkg.synthetic.make_segment(7, T, 32, U, n)
    cfg = pkg.CtcSegmentationParameters(index_duration=DUR)
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = pkg.ctc_segmentation.get_segments_device(cfg, [seg[0]], [seg[1]], [seg[2]])
        torch.cuda.synchronize()
        gpu = time.perf_counter() - t0
    t0 = time.perf_counter()
    o = oc.get_segments(*seg, oc.make_config(index_duration=DUR))
    cpu = time.perf_counter() - t0
    same = np.array_equal(res[0]["frame_of_label"], o["frame_of_label"])
    print(f"T={T} C={len(seg[1])}: GPU (host buffers, incl. alloc+copies) {gpu * 1e3:.0f} ms, oracle 1 core {cpu * 1e3:.0f} ms, "
          f"status {res[0]['status']}/{o['status']} frames equal: {same}", flush=True)

# batch throughput: windowed segments in flight at once (one workgroup each; a 9 500-frame window is
# bound by its sequential fp32 chain, so the chip only fills up across segments)
T, U, n = 9500, 40, 30
base = [pkg.synthetic.make_segment(20 + s, T, 32, U, n) for s in range(4)]
cfg = pkg.CtcSegmentationParameters(index_duration=DUR)
for B in (1, 32, 128, 256):
    segs = [base[i % 4] for i in range(B)]
    a, b, c = [s[0] for s in segs], [s[1] for s in segs], [s[2] for s in segs]
    pkg.ctc_segmentation.get_segments_device(cfg, a, b, c, want_state=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = pkg.ctc_segmentation.get_segments_device(cfg, a, b, c, want_state=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"T={T} x {B} windowed segments in one call: {dt * 1e3:.0f} ms = {B / dt:.1f} segments/s = "
          f"{B * T / dt:.3g} frames/s (host buffers, incl. copies)", flush=True)
