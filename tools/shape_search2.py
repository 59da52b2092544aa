#!/usr/bin/env python3
"""Tuning aid: for a set of (B, T, C, V) workloads, every compiled tile width K: fill / backtrack time
alone (serial schedule) and the pipelined step time, next to what the launch model picks (K=0).
    python tools/shape_search2.py [quick]"""
import os
import sys
import time
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


def run(segs, K, V, steps=24):
    T = [s[0].shape[0] for s in segs]
    C = [len(s[1]) for s in segs]
    U = [len(s[2]) - 1 for s in segs]
    try:
        plan = eng.plan(pkg.CtcSegmentationParameters(index_duration=DUR).to_native(), V, T, C, U, force_cols_per_lane=K)
    except Exception:
        return None
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).astype(dt)).to(dev)
    d_lpz = t(np.concatenate([s[0].reshape(-1) for s in segs]), np.float32)
    d_lab = t(np.concatenate([s[1] for s in segs]), np.int32)
    d_ub = t(np.concatenate([s[2] for s in segs]), np.int32)
    outs = [dict(fol=torch.empty(sum(C), dtype=torch.int32, device=dev), cp=torch.empty(sum(T), dtype=torch.float32, device=dev),
                 seg=torch.empty(3, max(1, sum(U)), dtype=torch.float64, device=dev),
                 te=torch.empty(len(segs), dtype=torch.int32, device=dev), st=torch.empty(len(segs), dtype=torch.int32, device=dev))
            for _ in range(3)]
    stream = torch.cuda.current_stream().cuda_stream
    n = [0]

    def step(pipelined):
        o = outs[n[0] % 3]
        n[0] += 1
        plan.run_device(d_lpz.data_ptr(), d_lab.data_ptr(), d_ub.data_ptr(), o["fol"].data_ptr(), o["cp"].data_ptr(), None,
                        o["seg"][0].data_ptr(), o["seg"][1].data_ptr(), o["seg"][2].data_ptr(), o["te"].data_ptr(),
                        o["st"].data_ptr(), stream, pipelined=pipelined)
    spin(lambda: step(False), ms=80.0)
    torch.cuda.synchronize()
    plan.set_timing(steps)
    for _ in range(steps):
        step(False)
    torch.cuda.synchronize()
    fill, bt = plan.get_timings(steps)
    plan.set_timing(0)
    for _ in range(8):
        step(True)
    plan.flush(stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4 * steps):
        step(True)
    plan.flush(stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (4 * steps)
    info = plan.info
    plan.close()
    return info["cols_per_lane"], info["waves_per_seg"], float(np.mean(fill)) * 1e3, float(np.mean(bt)) * 1e3, dt * 1e3


def uniform(B, T, U, n, V=32):
    base = [syn.make_segment(s, T, V, U, n) for s in range(8)]
    return [base[i % 8] for i in range(B)]


quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
work = [("B=512 C=640", 32, uniform(512, 3000, 22, 28)), ("B=512 C=640 V=38", 38, uniform(512, 3000, 22, 28, 38)),
        ("B=512 C=640 V=64", 64, uniform(512, 3000, 22, 28, 64)), ("B=1024 C=640", 32, uniform(1024, 3000, 22, 28)),
        ("B=512 C=256", 32, uniform(512, 3000, 6, 41)), ("B=512 C=1242", 32, uniform(512, 3000, 40, 30))]
if not quick:
    work += [("B=2048 C=640", 32, uniform(2048, 3000, 22, 28)), ("B=4096 C=640", 32, uniform(4096, 3000, 22, 28)),
             ("B=128 C=640", 32, uniform(128, 3000, 22, 28)), ("B=512 C=128", 32, uniform(512, 3000, 3, 41)),
             ("B=2048 C=256", 32, uniform(2048, 3000, 6, 41)), ("B=512 C=1536", 32, uniform(512, 3000, 59, 25)),
             ("B=128 C=1682 T=8000", 32, uniform(128, 8000, 60, 27)), ("B=4096 T=425 C=54", 32, uniform(4096, 425, 2, 25)),
             ("B=512 C=640 V=76", 76, uniform(512, 3000, 22, 28, 76)), ("B=512 C=640 V=128", 128, uniform(512, 3000, 22, 28, 128))]
for name, V, segs in work:
    out = []
    for K in (0, 1, 2, 3, 4, 5, 6, 8, 10, 12, 16):
        r = run(segs, K, V)
        if r:
            out.append(f"{'auto' if K == 0 else 'K'}{r[0]}/W{r[1]}: {r[2]:.0f}+{r[3]:.0f} step {r[4]:.3f}")
    print(name, "| " + " | ".join(out), flush=True)
