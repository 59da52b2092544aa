#!/usr/bin/env python3
"""Tuning aid: ONE small window (T frames, U utterances of n labels) through a plan, for each tile width K the library
compiles: fill alone and backtrack alone (HIP events, serial schedule).  usage: lone_k.py [T U n [B]]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as ge

pkg = ge.build()
a = [int(x) for x in sys.argv[1:]]
T, U, n = a[:3] if len(a) >= 3 else (499, 5, 20)
B = a[3] if len(a) > 3 else 1
syn = pkg.synthetic
lpz, gt, ub = syn.make_uniform_batch(B, T, 32, U, n)
C = gt.shape[1]
dev = torch.device("cuda:0")
d = [torch.from_numpy(x).to(dev) for x in (lpz.reshape(-1), gt.astype(np.int32).reshape(-1), ub.astype(np.int32).reshape(-1))]
o = dict(fol=torch.empty(B * C, dtype=torch.int32, device=dev), cp=torch.empty(B * T, dtype=torch.float32, device=dev),
         seg=torch.empty(3, B * U, dtype=torch.float64, device=dev), te=torch.empty(B, dtype=torch.int32, device=dev),
         st=torch.empty(B, dtype=torch.int32, device=dev))
stream = torch.cuda.current_stream().cuda_stream
cfg = pkg.CtcSegmentationParameters(index_duration=0.02)
eng = pkg._native.Engine(0)
for K in (0, 1, 2, 3, 4):
    try:
        plan = eng.plan(cfg.to_native(), 32, [T] * B, [C] * B, [U] * B, force_cols_per_lane=K)
    except Exception as e:
        print("K=%d: %s" % (K, e))
        continue
    run = lambda: plan.run_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), o["fol"].data_ptr(), o["cp"].data_ptr(), None,
                                  o["seg"][0].data_ptr(), o["seg"][1].data_ptr(), o["seg"][2].data_ptr(), o["te"].data_ptr(),
                                  o["st"].data_ptr(), stream)
    for _ in range(300):
        run()
    torch.cuda.synchronize()
    plan.set_timing(128)
    plan.set_timing_stride(1)
    for _ in range(128):
        run()
    torch.cuda.synchronize()
    f, b = plan.get_timings(128)
    print("T=%d C=%d B=%d K=%d (plan: K=%d W=%d): fill %.1f us (min %.1f), backtrack %.1f us" %
          (T, C, B, K, plan.info["cols_per_lane"], plan.info["waves_per_seg"], np.median(f) * 1e3, np.min(f) * 1e3, np.median(b) * 1e3), flush=True)
    plan.close()
