"""Diagnostic only: the TIMELINE a -DCTCFA_STAMP=4 build leaves behind (tools/build_variant.sh trace4 -DCTCFA_STAMP=4;
CTCFA_ALLOW_TUNING_BUILD=1 CTCFA_LIB=variants/trace4.so): an s_memtime stamp per tile and 16-row group, per producer and
block, for the first 16 workgroups, and each wave's HW_ID.  Prints, per workgroup: which SIMD every wave sits on, the
kernel's pace (cycles per group, by phase of the kernel), every tile's lag behind its left neighbour at each group end and how
often it must have waited, and what the producer's blocks cost.  usage: trace4.py [K] [B T U n] [--dump file.npz]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
pkg = ge.build()
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
if dump:
    argv = [a for a in argv if a != dump]
K = int(argv[0]) if argv else 0
syn = pkg.synthetic
B, T, V, U, n = 512, 3000, 32, 22, 28
if len(argv) > 4:
    B, T, U, n = (int(x) for x in argv[1:5])
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
d_cp = torch.zeros(max(B * T, 2 * (4 * 64 * 16 * 8 + 16 * 16 * 256)) + 64, dtype=torch.float32, device=dev)
d_seg = torch.zeros(3, B * U, dtype=torch.float64, device=dev)
d_te = torch.zeros(B, dtype=torch.int32, device=dev)
d_st = torch.zeros(B, dtype=torch.int32, device=dev)
for _ in range(300):
    plan.run_device(d_lpz.data_ptr(), d_lab.data_ptr(), d_ub.data_ptr(), d_fol.data_ptr(), d_cp.data_ptr(), None,
                    d_seg[0].data_ptr(), d_seg[1].data_ptr(), d_seg[2].data_ptr(), d_te.data_ptr(),
                    d_st.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
NW = min(B, 16)
base = 4 * 64 * 16 * 8
tr = d_cp.cpu().numpy().view(np.uint64)[base: base + NW * 16 * 256].reshape(NW, 16, 256).astype(np.int64)
if dump:
    np.savez_compressed(dump, trace=tr, W=W)
nblk = (T - 1 + 31) // 32
ng = 2 * nblk
for wg in range(min(NW, 4)):
    t = tr[wg]
    hw = t[:, 255]
    where = []
    for w in list(range(W)) + [14, 15]:
        if hw[w] == 0:
            continue
        h = int(hw[w]) & 0xffffffff
        where.append("%s:simd%d cu%d se%d xcc%d" % ("t%d" % w if w < 14 else "p%d" % (w - 14), (h >> 4) & 3, (h >> 8) & 15, (h >> 13) & 7, int(hw[w]) >> 32))
    print("== workgroup %d  %s" % (wg, "  ".join(where)))
    t0 = min(int(t[w, :ng][t[w, :ng] > 0].min()) for w in range(W) if (t[w, :ng] > 0).any())
    g_end = {w: t[w, :ng] for w in range(W)}
    # pace: cycles per group of each tile over thirds of its own active range
    for w in range(W):
        gs = np.nonzero(g_end[w] > 0)[0]
        if len(gs) < 8:
            continue
        d = np.diff(g_end[w][gs])
        same = np.diff(gs) == 1
        d = d[same]
        third = len(d) // 3
        ev = d[(gs[1:][same] % 2) == 0]   # groups that END mid-block (first half of a block)
        od = d[(gs[1:][same] % 2) == 1]   # groups that end a block (include the block boundary before them? no: after)
        print("  tile %d: groups %3d..%3d  first end +%7d  last end +%7d  cyc/group median %5.0f (first third %5.0f, middle %5.0f, last %5.0f)  "
              "1st-half groups %5.0f  2nd-half groups (+ block boundary) %5.0f  p90 %5.0f max %6d"
              % (w, gs[0], gs[-1], g_end[w][gs[0]] - t0, g_end[w][gs[-1]] - t0, np.median(d), np.median(d[:third]),
                 np.median(d[third:2 * third]), np.median(d[2 * third:]), np.median(ev), np.median(od), np.percentile(d, 90), d.max()))
    # lag of tile w behind tile w-1 at group ends: end_w(g) - end_{w-1}(g); a tile polls its neighbour ~4 rows before its own end
    for w in range(1, W):
        both = (g_end[w] > 0) & (g_end[w - 1] > 0)
        lag = (g_end[w] - g_end[w - 1])[both]
        if len(lag) == 0:
            continue
        per_row = np.median(np.diff(g_end[w][g_end[w] > 0])) / 16.0
        print("  tile %d behind tile %d at group ends: median %6.0f cyc (%4.1f rows), p10 %6.0f, p90 %6.0f; groups where it trails by < 8 rows: %3d of %3d"
              % (w, w - 1, np.median(lag), np.median(lag) / per_row, np.percentile(lag, 10), np.percentile(lag, 90),
                 int((lag < 8 * per_row).sum()), len(lag)))
    for p in (14, 15):
        pub = t[p, :nblk]
        if not (pub > 0).any():
            continue
        sp = t[p, 120:120 + min(nblk, 135)]
        n = min(len(sp), nblk)
        busy = pub[:n] - sp[:n]            # from "began to look for space" to "published": space wait + write
        gap = sp[1:n] - pub[:n - 1]        # publish -> next wait_space: the loads of the block after
        lead = []
        for jb in range(n):
            # blocks the producer is ahead of the slowest tile when it publishes jb: the group stamps tell which block each tile is in
            behind = [int((g_end[w][:ng] > 0).__and__(g_end[w][:ng] <= pub[jb]).sum()) // 2 for w in range(W) if (g_end[w] > 0).any()]
            lead.append(jb + 1 - min(behind))
        print("  producer %d: published block 0 at +%d, last at +%d; per block: wait-for-space + write median %5.0f (p90 %5.0f), in between %5.0f;"
              " blocks ahead of the slowest tile at publish: median %.0f, min %d"
              % (p - 14, pub[0] - t0, pub[n - 1] - t0, np.median(busy), np.percentile(busy, 90), np.median(gap), np.median(lead), min(lead)))
