#!/usr/bin/env python3
"""Tuning aid: wall time of ONE small alignment call (a 10 s / 60 s window), i.e. what an anchor
iteration pays per DP call: through the engine binding with host emissions, with device-resident
emissions, through the full aligner protocol, and 16 windows in one call."""
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
cs = pkg.ctc_segmentation


def timeit(fn, n):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n


for T, U, n in ((499, 4, 25), (2999, 22, 28)):
    seg = pkg.synthetic.make_segment(3, T, 32, U, n)
    dt = timeit(lambda: cs.get_segments_device(cfg, [seg[0]], [seg[1]], [seg[2]]), 300)
    print(f"T={T} C={len(seg[1])}: {dt * 1e6:.0f} us per call (host arrays in, results out)", flush=True)
    d = torch.from_numpy(seg[0]).cuda()
    dt = timeit(lambda: cs.get_segments_device(cfg, [d], [seg[1]], [seg[2]]), 300)
    print(f"T={T} C={len(seg[1])}: {dt * 1e6:.0f} us per call (emissions resident in HBM)", flush=True)
    eng = cs.default_engine()
    prm = cfg.to_native()
    lab = [np.ascontiguousarray(seg[1], np.int32)]
    dt = timeit(lambda: eng.align_batch(prm, [seg[0]], lab, [seg[2]]), 300)
    print(f"T={T} C={len(seg[1])}: {dt * 1e6:.0f} us per call (engine binding only, host arrays)", flush=True)
    segs = [pkg.synthetic.make_segment(10 + i, T, 32, U, n) for i in range(16)]
    a, b, c = [s[0] for s in segs], [s[1] for s in segs], [s[2] for s in segs]
    dt = timeit(lambda: cs.get_segments_device(cfg, a, b, c), 50)
    print(f"T={T} x16 windows: {dt * 1e6:.0f} us per call = {dt / 16 * 1e6:.0f} us per window", flush=True)

# kernel times of the same single-window launches (HIP events on the kernels' dispatch packets)
for T, U, n in ((499, 4, 25), (2999, 22, 28)):
    seg = pkg.synthetic.make_segment(3, T, 32, U, n)
    eng = cs.default_engine()
    plan = eng.plan(cfg.to_native(), 32, [T], [len(seg[1])], [len(seg[2]) - 1])
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
    step = lambda: plan.run_device(d_lpz.data_ptr(), d_lab.data_ptr(), d_ub.data_ptr(), fol.data_ptr(), cp.data_ptr(), None,
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
    print(f"T={T} C={len(seg[1])}: kernels fill {np.median(f) * 1e3:.1f} us + backtrack {np.median(b) * 1e3:.1f} us "
          f"(K={info['cols_per_lane']} W={info['waves_per_seg']})", flush=True)
