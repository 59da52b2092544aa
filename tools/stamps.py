"""Diagnostic only: read the s_memtime stamps an instrumented build (variants/exp_stamp.so,
built from a throw-away copy of the kernel with four stamps per step) leaves in the
char_prob buffer, and print where a pipeline step's cycles go."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import __graft_entry__ as ge  # noqa: E402

pkg = ge.build()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
syn = pkg.synthetic
B, T, V, U, n = 512, 3000, 32, 22, 28
lpz, gt, ub = syn.make_uniform_batch(B, T, V, U, n)
C = gt.shape[1]
cfg = pkg.CtcSegmentationParameters(index_duration=0.02)
eng = pkg._native.Engine(0)
plan = eng.plan(cfg.to_native(), V, [T] * B, [C] * B, [U] * B, force_cols_per_lane=K)
W = plan.info["waves_per_seg"]
dev = torch.device("cuda:0")
d_lpz = torch.from_numpy(lpz.reshape(-1)).to(dev)
d_lab = torch.from_numpy(gt.astype(np.int32).reshape(-1)).to(dev)
d_ub = torch.from_numpy(ub.astype(np.int32).reshape(-1)).to(dev)
d_fol = torch.zeros(B * C, dtype=torch.int32, device=dev)
d_cp = torch.zeros(B * T, dtype=torch.float32, device=dev)
d_seg = torch.zeros(3, B * U, dtype=torch.float64, device=dev)
d_te = torch.zeros(B, dtype=torch.int32, device=dev)
d_st = torch.zeros(B, dtype=torch.int32, device=dev)
for _ in range(3):
    plan.run_device(d_lpz.data_ptr(), d_lab.data_ptr(), d_ub.data_ptr(), d_fol.data_ptr(), d_cp.data_ptr(), None,
                    d_seg[0].data_ptr(), d_seg[1].data_ptr(), d_seg[2].data_ptr(), d_te.data_ptr(),
                    d_st.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
raw = d_cp.cpu().numpy().reshape(B, T)
nblk = (T - 1 + 31) // 32
nsteps = nblk + W - 1
for seg in range(2):
    st = raw[seg].view(np.uint64)
    n = min(len(st) // 6, W * nsteps)
    st = st[: n * 6].reshape(-1, 6).astype(np.int64)
    for w in range(W):
        rows = st[w * nsteps:(w + 1) * nsteps]
        rows = rows[(rows > 0).all(axis=1)]
        if len(rows) < 30:
            continue
        mid = rows[10:-10]
        d = [np.median(mid[:, i + 1] - mid[:, i]) for i in range(5)]
        step = np.median(mid[1:, 0] - mid[:-1, 0])
        print(f"seg {seg} wave {w}: step {step:.0f} cyc = store_dec+stage_load {d[0]:.0f} + compute {d[1]:.0f} "
              f"+ vmcnt_wait {d[2]:.0f} + stage_write {d[3]:.0f} + barrier {d[4]:.0f}  (K={K}, W={W})")
