#!/usr/bin/env python3
"""Tuning aid: fill time for every compiled tile width on a few (B, C) workloads, next to the
width the launch heuristic picks (K=0)."""
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


def run(segs, K, steps=30):
    T = [s[0].shape[0] for s in segs]
    C = [len(s[1]) for s in segs]
    U = [len(s[2]) - 1 for s in segs]
    try:
        plan = eng.plan(pkg.CtcSegmentationParameters(index_duration=DUR).to_native(), 32, T, C, U, force_cols_per_lane=K)
    except Exception:
        return None
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
    step = lambda: plan.run_device(d_lpz.data_ptr(), d_lab.data_ptr(), d_ub.data_ptr(), fol.data_ptr(), cp.data_ptr(), None,
                                   seg[0].data_ptr(), seg[1].data_ptr(), seg[2].data_ptr(), te.data_ptr(), st.data_ptr(), stream)
    spin(step)
    torch.cuda.synchronize()
    plan.set_timing(steps)
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    fill, bt = plan.get_timings(steps)
    info = plan.info
    plan.close()
    return info["cols_per_lane"], info["waves_per_seg"], float(np.mean(fill)) * 1e3


def uniform(B, T, U, n):
    base = [syn.make_segment(s, T, 32, U, n) for s in range(8)]
    return [base[i % 8] for i in range(B)]


for name, segs in (("B=4096 C=640", uniform(4096, 3000, 22, 28)), ("B=2048 C=640", uniform(2048, 3000, 22, 28)),
                   ("B=1024 C=640", uniform(1024, 3000, 22, 28)), ("B=256 C=640", uniform(256, 3000, 22, 28)),
                   ("B=512 C=1536", uniform(512, 3000, 59, 25)), ("B=512 C=256", uniform(512, 3000, 6, 41)),
                   ("B=2048 C=256", uniform(2048, 3000, 6, 41)), ("B=128 C=1682 T=8000", uniform(128, 8000, 60, 27))):
    out = []
    for K in (0, 1, 2, 3, 4, 5, 6, 8, 10, 12, 16):
        r = run(segs, K)
        if r:
            out.append(f"{'auto' if K == 0 else 'K'}{r[0]}/W{r[1]}:{r[2]:.0f}")
    print(name, " ".join(out), flush=True)
