import os, sys, numpy as np
sys.path.insert(0, "/root/repo")
import torch
import __graft_entry__ as ge
pkg = ge.build()
cs = pkg.ctc_segmentation
cfg = pkg.CtcSegmentationParameters(index_duration=0.02)
eng = cs.default_engine()
dev = torch.device("cuda:0")
for T, U, n, B in ((499, 4, 25, 1), (499, 4, 25, 16), (499, 4, 25, 64), (2999, 22, 28, 1), (2999, 22, 28, 16), (300, 3, 12, 1), (300, 3, 12, 64)):
    seg = pkg.synthetic.make_segment(3, T, 32, U, n)
    C = len(seg[1])
    out = []
    for K in (0, 1, 2, 3, 4, 5, 6, 8):
        try:
            plan = eng.plan(cfg.to_native(), 32, [T] * B, [C] * B, [len(seg[2]) - 1] * B, force_cols_per_lane=K)
        except Exception:
            continue
        d_lpz = torch.from_numpy(np.tile(seg[0].reshape(-1), B)).to(dev)
        d_lab = torch.from_numpy(np.tile(seg[1].astype(np.int32), B)).to(dev)
        d_ub = torch.from_numpy(np.tile(seg[2].astype(np.int32), B)).to(dev)
        fol = torch.empty(C * B, dtype=torch.int32, device=dev)
        cp = torch.empty(T * B, dtype=torch.float32, device=dev)
        sg = torch.empty(3, (len(seg[2]) - 1) * B, dtype=torch.float64, device=dev)
        te = torch.empty(B, dtype=torch.int32, device=dev)
        st = torch.empty(B, dtype=torch.int32, device=dev)
        stream = torch.cuda.current_stream().cuda_stream
        step = lambda: plan.run_device(d_lpz.data_ptr(), d_lab.data_ptr(), d_ub.data_ptr(), fol.data_ptr(), cp.data_ptr(), None,
                                       sg[0].data_ptr(), sg[1].data_ptr(), sg[2].data_ptr(), te.data_ptr(), st.data_ptr(), stream)
        for _ in range(100):
            step()
        torch.cuda.synchronize()
        plan.set_timing(30)
        for _ in range(30):
            step()
            torch.cuda.synchronize()
        f, b = plan.get_timings(30)
        info = plan.info
        out.append(f"{'auto' if K == 0 else 'K'}{info['cols_per_lane']}/W{info['waves_per_seg']}: {np.median(f) * 1e3:.1f}+{np.median(b) * 1e3:.1f}")
        plan.close()
    print(f"B={B} T={T} C={C} |", " | ".join(out), flush=True)
