#!/usr/bin/env python3
"""Throughput of the HIP path over the shape sweeps SURVEY.md §8(d) lists (C, T, B, ragged T,
word-level rows).  Serial schedule, device-resident inputs, HIP-event kernel times.
Prints a markdown table (copied into profiles/)."""
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


def run(name, segs, steps=10, warm=3):
    T = [s[0].shape[0] for s in segs]
    C = [len(s[1]) for s in segs]
    U = [len(s[2]) - 1 for s in segs]
    V = segs[0][0].shape[1]
    plan = eng.plan(pkg.CtcSegmentationParameters(index_duration=DUR).to_native(), V, T, C, U)
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
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    fill, bt = plan.get_timings(steps)
    assert (st.cpu().numpy() == 0).all()
    frames = sum(T)
    info = plan.info
    print(f"| {name} | {len(segs)} | {min(T)}-{max(T)} | {min(C)}-{max(C)} | K={info['cols_per_lane']} W={info['waves_per_seg']} | "
          f"{np.mean(fill) * 1e3:.0f} | {np.mean(bt) * 1e3:.0f} | {dt * 1e3:.3f} | {frames / dt:.3g} | "
          f"{info['algorithmic_bytes'] / (np.mean(fill) * 1e-3) / 1e9:.0f} |")
    plan.close()


def uniform(B, T, U, n, V=32, seed0=0):
    base = [syn.make_segment(seed0 + s, T, V, U, n) for s in range(min(B, 16))]
    return [base[i % len(base)] for i in range(B)]


print("| workload | B | T | C | shape | fill us | backtrack us | ms/step (serial) | frames/s | fill GB/s (algorithmic) |")
print("|---|---|---|---|---|---|---|---|---|---|")
run("config 3 (C=640)", uniform(512, 3000, 22, 28))
run("C=256", uniform(512, 3000, 6, 41))
run("C=1536", uniform(512, 3000, 59, 25))
run("T=500", uniform(512, 500, 4, 24))
run("T=8000", uniform(128, 8000, 60, 27))
run("B=64", uniform(64, 3000, 22, 28))
run("B=4096", uniform(4096, 3000, 22, 28))
rng = np.random.default_rng(0)
rag = []
for s in range(512):
    T = int(rng.integers(150, 3500))
    U = max(1, T // 140)
    rag.append(syn.make_segment(5000 + s % 32, T, 32, U, 24))
run("ragged T~U[150,3500]", rag)
rag4 = []
for s in range(2048):
    T = int(rng.integers(150, 3500))
    U = max(1, T // 140)
    rag4.append(syn.make_segment(5000 + s % 32, T, 32, U, 24))
run("ragged T~U[150,3500], 2048 segments", rag4)
words = []
for s in range(4096):
    T = int(rng.integers(100, 750))
    U = int(rng.choice([3, 5]))
    words.append(syn.make_segment(7000 + s % 64, T, 32, U, max(2, min(12, (T - 8) // (U * 2)))))
run("word-level rows (config 4 shape)", words)
run("one 10 s window (config 1 shape)", uniform(1, 499, 3, 34))
