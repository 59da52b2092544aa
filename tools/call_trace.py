#!/usr/bin/env python3
"""Tuning aid: host time of one small call by phase (CTCFA_CALL_TRACE=1 makes the library print it when
the engine goes away) -- engine binding only, host emissions and resident emissions."""
import os
import sys
import time
os.environ["CTCFA_CALL_TRACE"] = "1"
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

pkg = ge.build()
cfg = pkg.CtcSegmentationParameters(index_duration=320.4769 / 16000)
prm = cfg.to_native()
for T, U, n in ((499, 4, 25), (2999, 22, 28)):
    seg = pkg.synthetic.make_segment(3, T, 32, U, n)
    lab = [np.ascontiguousarray(seg[1], np.int32)]
    d = torch.from_numpy(seg[0]).cuda()
    for resident in (False, True):
        eng = pkg._native.Engine(0)
        kw = dict(d_lpz=d.data_ptr(), stream=torch.cuda.current_stream().cuda_stream, shapes=[seg[0].shape]) if resident else {}
        fn = lambda: eng.align_batch(prm, None if resident else [seg[0]], lab, [seg[2]], **kw)
        for _ in range(200):
            fn()
        t0 = time.perf_counter()
        for _ in range(500):
            fn()
        dt = (time.perf_counter() - t0) / 500
        print(f"T={T} resident={resident}: {dt * 1e6:.1f} us per call through the binding", flush=True)
        sys.stdout.flush()
        eng.close()
