"""Diagnostic only: per-wave cycle totals a -DCTCFA_BT_STAMP build of the checkpoint-mode backtrack
(stride_backtrack_kernel) leaves in the caller's `state` buffer.
    tools/build_variant.sh btstamp -DCTCFA_BT_STAMP && CTCFA_ALLOW_TUNING_BUILD=1 CTCFA_LIB=$PWD/variants/btstamp.so python tools/bt_stamps.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
pkg = ge.build()
syn = pkg.synthetic
B, T, V, U, n = 512, 3000, 32, 22, 28
if len(sys.argv) > 4:
    B, T, U, n = (int(x) for x in sys.argv[1:5])
lpz, gt, ub = syn.make_uniform_batch(B, T, V, U, n)
C = gt.shape[1]
cfg = pkg.CtcSegmentationParameters(index_duration=0.02)
cfg.backtrack_from_max_t = bool(int(os.environ.get("FROM_MAX_T", "0")))
eng = pkg._native.Engine(0)
plan = eng.plan(cfg.to_native(), V, [T] * B, [C] * B, [U] * B)
dev = torch.device("cuda:0")
d_lpz = torch.from_numpy(lpz.reshape(-1)).to(dev)
d_lab = torch.from_numpy(gt.astype(np.int32).reshape(-1)).to(dev)
d_ub = torch.from_numpy(ub.astype(np.int32).reshape(-1)).to(dev)
d_fol = torch.zeros(B * C, dtype=torch.int32, device=dev)
d_cp = torch.zeros(B * T, dtype=torch.float32, device=dev)
d_state = torch.zeros(B * T, dtype=torch.int32, device=dev)
d_seg = torch.zeros(3, B * U, dtype=torch.float64, device=dev)
d_te = torch.zeros(B, dtype=torch.int32, device=dev)
d_st = torch.zeros(B, dtype=torch.int32, device=dev)
for _ in range(300):
    plan.run_device(d_lpz.data_ptr(), d_lab.data_ptr(), d_ub.data_ptr(), d_fol.data_ptr(), d_cp.data_ptr(), d_state.data_ptr(),
                    d_seg[0].data_ptr(), d_seg[1].data_ptr(), d_seg[2].data_ptr(), d_te.data_ptr(),
                    d_st.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
te = d_te.cpu().numpy()
raw = d_state.cpu().numpy().reshape(B, T)[:, :8 * 16 * 2].copy().view(np.uint64).reshape(B, 8, 16).astype(np.int64)
nw = int((raw[0, :, 0] != 0).sum())
print(f"t_end median {np.median(te):.0f}  blocks {np.median((te - 1) // 32 + 1):.0f}  waves {nw}")
w0 = raw[:, 0, :]
print(f"kernel (wave 0): end cell known +{np.median(w0[:,1]):.0f}  chain done +{np.median(w0[:,2]):.0f}  "
      f"barrier +{np.median(w0[:,3]):.0f}  scored +{np.median(w0[:,4]):.0f} cycles")
names = {5: "first rows", 6: "anchor", 7: "recompute", 8: "entry wait", 9: "walk", 10: "publish+frames", 11: "slot refill"}
for w in range(nw):
    r = raw[:, w, :]
    turns = np.maximum(np.median(r[:, 13]), 1)
    print(f"wave {w}: turns {turns:.0f} fallbacks {np.median(r[:,14]):.1f} chain done +{np.median(r[:,2]):.0f} | per turn: " +
          "  ".join(f"{names[k]} {np.median(r[:,k]) / (1 if k == 5 else turns):.0f}" for k in (5, 6, 7, 8, 9, 10, 11)) +
          f"  | extra windows {np.median(r[:,12]):.0f}")
