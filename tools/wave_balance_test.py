#!/usr/bin/env python3
"""Tuning aid: fill-kernel time vs number of K=2 compute waves per workgroup (C = 128*W).
If a step is bound by the most crowded SIMD, W=5 (3,3,2,2 compute waves per SIMD) pays for 3
while W=4 and W=3 pay for 2 -- and W=3 costs the same as W=4 (per-wave issue bound)."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from _spinup import spin  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

pkg = ge.build()
syn = pkg.synthetic
DUR = 320.4769 / 16000
dev = torch.device("cuda:0")
eng = pkg._native.Engine(0)


def run(name, segs, K, steps=10, warm=3):
    T = [s[0].shape[0] for s in segs]
    C = [len(s[1]) for s in segs]
    U = [len(s[2]) - 1 for s in segs]
    V = segs[0][0].shape[1]
    plan = eng.plan(pkg.CtcSegmentationParameters(index_duration=DUR).to_native(), V, T, C, U, force_cols_per_lane=K)
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).astype(dt)).to(dev)
    d_lpz = t(np.concatenate([s[0].reshape(-1) for s in segs]), np.float32)
    d_lab = t(np.concatenate([s[1] for s in segs]), np.int32)
    d_ub = t(np.concatenate([s[2] for s in segs]), np.int32)
    fol = torch.empty(sum(C), dtype=torch.int32, device=dev)
    cp = torch.empty(sum(T), dtype=torch.float32, device=dev)
    seg = torch.empty(3, max(1, sum(U)), dtype=torch.float64, device=dev)
    te = torch.empty(len(segs), dtype=torch.int32, device=dev)
    st = torch.empty(len(segs), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        plan.run_device(d_lpz.data_ptr(), d_lab.data_ptr(), d_ub.data_ptr(), fol.data_ptr(), cp.data_ptr(), None,
                        seg[0].data_ptr(), seg[1].data_ptr(), seg[2].data_ptr(), te.data_ptr(), st.data_ptr(), stream)
    spin(step)
    torch.cuda.synchronize()
    plan.set_timing(steps)
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    fill, bt = plan.get_timings(steps)
    info = plan.info
    print(f"{name}: C={max(C)} K={info['cols_per_lane']} W={info['waves_per_seg']} fill {np.mean(fill) * 1e3:.0f} us  bt {np.mean(bt) * 1e3:.0f} us", flush=True)
    plan.close()


def uniform(B, T, U, n, V=32):
    base = [syn.make_segment(s, T, V, U, n) for s in range(8)]
    return [base[i % len(base)] for i in range(B)]


# C = 1 + U*(1+n) + 1
for U, n, K in ((22, 28, 2), (17, 29, 2), (14, 26, 2), (13, 19, 2), (9, 27, 2), (22, 28, 1), (10, 30, 1), (5, 24, 1)):
    run("B=512 T=3000", uniform(512, 3000, U, n), K)
