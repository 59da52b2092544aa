"""Diagnostic only: cycles per phase of the backtrack kernel (build with -DCTCFA_BT_PHASES)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
pkg = ge.build()
syn = pkg.synthetic
B, T, V, U, n = 512, 3000, 32, 22, 28
lpz, gt, ub = syn.make_uniform_batch(B, T, V, U, n)
C = gt.shape[1]
eng = pkg._native.Engine(0)
plan = eng.plan(pkg.CtcSegmentationParameters(index_duration=0.02).to_native(), V, [T] * B, [C] * B, [U] * B)
dev = torch.device("cuda:0")
t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a.reshape(-1)).astype(dt)).to(dev)
d_lpz, d_lab, d_ub = t(lpz, np.float32), t(gt, np.int32), t(ub, np.int32)
fol = torch.zeros(B * C, dtype=torch.int32, device=dev); cp = torch.zeros(B * T, dtype=torch.float32, device=dev)
seg = torch.zeros(3, B * U, dtype=torch.float64, device=dev); te = torch.zeros(B, dtype=torch.int32, device=dev); st = torch.zeros(B, dtype=torch.int32, device=dev)
for _ in range(300):
    plan.run_device(d_lpz.data_ptr(), d_lab.data_ptr(), d_ub.data_ptr(), fol.data_ptr(), cp.data_ptr(), None, seg[0].data_ptr(),
                    seg[1].data_ptr(), seg[2].data_ptr(), te.data_ptr(), st.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
s = seg[0].cpu().numpy().reshape(B, U)
for q, name in enumerate(("end cell + staging", "walk", "per-frame outputs", "utterance scores")):
    print("  %-20s %7.0f ticks (median over segments)" % (name, np.median(s[:, q])))
