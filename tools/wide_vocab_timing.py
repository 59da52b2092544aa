#!/usr/bin/env python3
"""Tuning aid: anchor-window-shaped batches over a wide (sub-word) vocabulary through the host-buffer / resident
entries: the compact-matrix path (default) against the gather kernel (CTCFA_NO_REMAP=1)."""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

pkg = ge.build()
cfg = pkg.CtcSegmentationParameters(index_duration=320.4769 / 16000)
cs = pkg.ctc_segmentation
for V, B, T, U, n in ((256, 256, 500, 4, 20), (1000, 256, 500, 4, 20), (5000, 64, 500, 4, 20), (1000, 64, 1500, 6, 18)):
    rng = np.random.default_rng(V)
    segs = []
    for b in range(8):
        gt, ub = pkg.synthetic.make_labels(rng, U, n, V)
        segs.append((pkg.synthetic.make_emissions(rng, T, V, gt), gt, ub))
    segs = [segs[i % 8] for i in range(B)]
    dev = [torch.from_numpy(s[0]).cuda() for s in segs[:8]]
    lpz = [dev[i % 8].clone() for i in range(B)]   # (distinct tensors: no shared emissions)
    for mode in ("compact", "gather"):
        os.environ.pop("CTCFA_NO_REMAP", None)
        if mode == "gather":
            os.environ["CTCFA_NO_REMAP"] = "1"
        run = lambda: cs.get_segments_device(cfg, lpz, [s[1] for s in segs], [s[2] for s in segs])
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            run()
        dt = (time.perf_counter() - t0) / 10
        print(f"V={V} B={B} T={T} C={len(segs[0][1])}: {mode:8s} {dt * 1e3:.2f} ms per call = {B * T / dt:.3g} frames/s", flush=True)
