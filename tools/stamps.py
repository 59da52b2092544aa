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
K = int(sys.argv[1]) if len(sys.argv) > 1 else 0   # 0: the automatic choice (mixed 8-wave shape for config 3)
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
for seg in range(1):
    NW = 8 if K == 0 else W + 1
    roles = {0: "heavy s0", 2: "heavy s1", 4: "heavy s2", 6: "heavy s3", 1: "light s4", 3: "light s5", 5: "producer", 7: "idle"} \
        if K == 0 else {w: ("producer" if w == W else f"stage {w}") for w in range(W + 1)}
    st = raw[seg].view(np.uint64)[: NW * 16 * 3].reshape(NW, 16, 3).astype(np.int64)
    t_ref = st[0, 0, 0]
    for w in range(NW):
        rows = st[w]
        ok = (rows > 0).all(axis=1)
        if ok.sum() < 8:
            print(f"wave {w}: no stamps")
            continue
        rows = rows[ok]
        work = np.median(rows[:, 1] - rows[:, 0])
        bar = np.median(rows[:, 2] - rows[:, 1])
        step = np.median(rows[1:, 0] - rows[:-1, 0])
        role = roles[w]
        print(f"wave {w:2d} {role:9s}: step {step:.0f} cyc = work {work:.0f} + barrier {bar:.0f}   "
              f"(work ends at +{np.median(rows[:, 1] - st[0, ok, 0]):.0f} after wave 0 starts the step)  K={K} W={W}")
