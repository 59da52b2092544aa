"""Diagnostic only: per-tile cycle totals an instrumented build (-DCTCFA_STAMP) leaves in the lastcol
workspace: total, waiting for the left neighbour, waiting for staged emissions, and (CTCFA_STAMP=3) a stamp
every 8 rows of block 40.  Such a build passes the caller's char_prob buffer as the stamp buffer and skips
the backtrack (tools/build_variant.sh stamp3 -DCTCFA_STAMP=3; CTCFA_ALLOW_TUNING_BUILD=1 CTCFA_LIB=variants/stamp3.so)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
pkg = ge.build()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 0
syn = pkg.synthetic
B, T, V, U, n = 512, 3000, 32, 22, 28
if len(sys.argv) > 5:
    B, T, U, n = (int(x) for x in sys.argv[2:6])
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
d_cp = torch.zeros(B * T + (B * T) % 2 + 3 * 64 * 16 * 8 * 2, dtype=torch.float32, device=dev)   # (room for the stamps of short windows)
d_seg = torch.zeros(3, B * U, dtype=torch.float64, device=dev)
d_te = torch.zeros(B, dtype=torch.int32, device=dev)
d_st = torch.zeros(B, dtype=torch.int32, device=dev)
for _ in range(300):
    plan.run_device(d_lpz.data_ptr(), d_lab.data_ptr(), d_ub.data_ptr(), d_fol.data_ptr(), d_cp.data_ptr(), None,
                    d_seg[0].data_ptr(), d_seg[1].data_ptr(), d_seg[2].data_ptr(), d_te.data_ptr(),
                    d_st.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
nb = min(B, 64)
raw = d_cp.cpu().numpy().view(np.uint64)[: nb * 16 * 8].reshape(nb, 16, 8).astype(np.int64)
for w in range(W):
    r = raw[:, w, :]
    print(f"tile {w}: total {np.median(r[:,0]):.0f} cyc  nbr-wait {np.median(r[:,1]):.0f} ({np.median(r[:,3]):.0f} x)  "
          f"staged-wait {np.median(r[:,2]):.0f} ({np.median(r[:,4]):.0f} x)  first two blocks done after {np.median(r[:,5] >> 32):.0f} / {np.median(r[:,5] & 0xffffffff):.0f}  last block {np.median(r[:,6]):.0f}  "
          f"start +{np.median(r[:,7]-raw[:,0,7]):.0f}")
r = raw[:, 15, :]
print(f"producer: total {np.median(r[:,0]):.0f} cyc  space-wait {np.median(r[:,1]):.0f} ({np.median(r[:,3]):.0f} x)  load-wait + write {np.median(r[:,2]):.0f}  blocks {np.median(r[:,4]):.0f}")
SB = int(os.environ.get("STAMP_BLOCK", "40"))   # (a build with -DCTCFA_STAMP_BLOCK=<n> for short windows)
if T <= (SB + 2) * 32:
    sys.exit(0)
u64 = d_cp.cpu().numpy().view(np.uint64)
first = u64[64 * 16 * 8: 64 * 16 * 8 + nb * 16 * 8].reshape(nb, 16, 8).astype(np.int64)
second = u64[2 * 64 * 16 * 8: 2 * 64 * 16 * 8 + nb * 16 * 8].reshape(nb, 16, 8).astype(np.int64)
for w in range(W):
    q, q2 = first[:, w, :5], second[:, w, :5]
    print(f"tile {w} block {SB}: cycles per 8 rows", [int(np.median(q[:, i + 1] - q[:, i])) for i in range(4)],
          f" to the first row of block {SB + 1}: {int(np.median(q2[:, 0] - q[:, 4]))}",
          f" block {SB + 1}:", [int(np.median(q2[:, i + 1] - q2[:, i])) for i in range(4)])
