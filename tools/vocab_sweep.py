#!/usr/bin/env python3
"""Tuning aid: config-3-shaped batch (512 x 3000 frames, C = 640) over vocabulary sizes -- the
reference's Spanish wav2vec2 model has a 38-token character vocabulary."""
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
# texts: labels drawn from the whole vocabulary (V - 1 of them: no ring takes those above 32 / 64), from 28 entries of it (a
# character model's window: the 32-entry ring), from 50 (the 64-entry ring above 64 entries).  Plans are created with
# their labels, as the host-buffer entries do.
for V, alphabet in [(V, a) for V in (20, 29, 32, 38, 48, 64, 76, 100, 128, 160, 192, 256, 300) for a in (None, 28, 50)
                    if a is None or (a == 28 and 32 < V <= 256) or (a == 50 and 64 < V <= 256)]:
    base = [syn.make_segment(s, 3000, V, 22, 28, alphabet=alphabet) for s in range(8)]
    segs = [base[i % 8] for i in range(512)]
    T = [s[0].shape[0] for s in segs]
    C = [len(s[1]) for s in segs]
    U = [len(s[2]) - 1 for s in segs]
    plan = eng.plan(pkg.CtcSegmentationParameters(index_duration=DUR).to_native(), V, T, C, U,
                    labels=np.concatenate([s[1] for s in segs]).astype(np.int32) if V <= 256 else None)
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).astype(dt)).to(dev)
    d_lpz = t(np.concatenate([s[0].reshape(-1) for s in segs]), np.float32)
    d_lab = t(np.concatenate([s[1] for s in segs]), np.int32)
    d_ub = t(np.concatenate([s[2] for s in segs]), np.int32)
    fol = torch.empty(sum(C), dtype=torch.int32, device=dev)
    cp = torch.empty(sum(T), dtype=torch.float32, device=dev)
    seg = torch.empty(3, sum(U), dtype=torch.float64, device=dev)
    te = torch.empty(512, dtype=torch.int32, device=dev)
    st = torch.empty(512, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    step = lambda: plan.run_device(d_lpz.data_ptr(), d_lab.data_ptr(), d_ub.data_ptr(), fol.data_ptr(), cp.data_ptr(), None,
                                   seg[0].data_ptr(), seg[1].data_ptr(), seg[2].data_ptr(), te.data_ptr(), st.data_ptr(), stream)
    spin(step)
    torch.cuda.synchronize()
    plan.set_timing(10)
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    fill, bt = plan.get_timings(10)
    info = plan.info
    print(f"V={V} texts over {alphabet or V - 1} entries: K={info['cols_per_lane']} stages={info['waves_per_seg']} pitch={info['vocab_pitch']} lds={info['lds_bytes']} "
          f"fill {np.mean(fill) * 1e3:.0f} us bt {np.mean(bt) * 1e3:.0f} us", flush=True)
    plan.close()
