"""Diagnostic only: the TIMELINE a -DCTCFA_STAMP=4 build leaves behind (tools/build_variant.sh trace4 -DCTCFA_STAMP=4;
CTCFA_ALLOW_TUNING_BUILD=1 CTCFA_LIB=variants/trace4.so): s_memtime stamps per tile -- [g] end of 16-row group g, [256+g]
arrival at the neighbour poll of group g, [512+g] end of a wait there, [768+j] start of block j -- per producer and block
([jb] published, [120+jb] began to look for ring space), for the first 16 workgroups, and each wave's HW_ID.
Prints, per workgroup: where the waves sit, each tile's pace and what its groups are made of, the lag between neighbours,
how long the polls waited, the producer's blocks.  usage: trace4.py [K] [B T U n] [--dump file.npz] [--timeline wg]
CTCFA_TRACE_V=38: a vocabulary of that many entries, texts over 28 of them, through a narrowed plan."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
pkg = ge.build()
raw = sys.argv[1:]
dump = raw[raw.index("--dump") + 1] if "--dump" in raw else None
tl_wg = int(raw[raw.index("--timeline") + 1]) if "--timeline" in raw else -1
argv = []
skip = False
for a in raw:
    if skip:
        skip = False
        continue
    if a in ("--dump", "--timeline"):
        skip = True
        continue
    argv.append(a)
K = int(argv[0]) if argv else 0
syn = pkg.synthetic
B, T, V, U, n = 512, 3000, int(os.environ.get("CTCFA_TRACE_V", "32")), 22, 28
if len(argv) > 4:
    B, T, U, n = (int(x) for x in argv[1:5])
lpz, gt, ub = syn.make_uniform_batch(B, T, V, U, n, alphabet=28 if V > 32 else None)
C = gt.shape[1]
cfg = pkg.CtcSegmentationParameters(index_duration=0.02)
eng = pkg._native.Engine(0)
plan = eng.plan(cfg.to_native(), V, [T] * B, [C] * B, [U] * B, force_cols_per_lane=K, texts_of_31_labels=V > 32)
W = plan.info["waves_per_seg"]
dev = torch.device("cuda:0")
d_lpz = torch.from_numpy(lpz.reshape(-1)).to(dev)
d_lab = torch.from_numpy(gt.astype(np.int32).reshape(-1)).to(dev)
d_ub = torch.from_numpy(ub.astype(np.int32).reshape(-1)).to(dev)
d_fol = torch.zeros(B * C, dtype=torch.int32, device=dev)
SL = 1024
TB = (B * T + 1) // 2 * 2 + 4          # floats: the timeline starts 16 bytes past the end of char_prob (rounded to 8 bytes)
d_cp = torch.zeros(TB + 2 * 16 * 16 * SL + 64, dtype=torch.float32, device=dev)
d_seg = torch.zeros(3, B * U, dtype=torch.float64, device=dev)
d_te = torch.zeros(B, dtype=torch.int32, device=dev)
d_st = torch.zeros(B, dtype=torch.int32, device=dev)
for _ in range(300):
    plan.run_device(d_lpz.data_ptr(), d_lab.data_ptr(), d_ub.data_ptr(), d_fol.data_ptr(), d_cp.data_ptr(), None,
                    d_seg[0].data_ptr(), d_seg[1].data_ptr(), d_seg[2].data_ptr(), d_te.data_ptr(),
                    d_st.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
d_cp[TB:].zero_()   # one more launch into a clean timeline (a wait stamp is only written when a wait happened)
plan.run_device(d_lpz.data_ptr(), d_lab.data_ptr(), d_ub.data_ptr(), d_fol.data_ptr(), d_cp.data_ptr(), None,
                d_seg[0].data_ptr(), d_seg[1].data_ptr(), d_seg[2].data_ptr(), d_te.data_ptr(),
                d_st.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
NW = min(B, 16)
base = TB // 2
tr = d_cp.cpu().numpy().view(np.uint64)[base: base + NW * 16 * SL].reshape(NW, 16, SL).astype(np.int64)
if dump:
    np.savez_compressed(dump, trace=tr, W=W)
nblk = (T - 1 + 31) // 32
ng = 2 * nblk
med = lambda x: float(np.median(x)) if len(x) else float("nan")
for wg in ([tl_wg] if tl_wg >= 0 else range(min(NW, 4))):
    t = tr[wg]
    where = []
    for w in list(range(W)) + [14, 15]:
        hwx = int(t[w, SL - 1])
        if hwx == 0:
            continue
        h = hwx & 0xffffffff
        where.append("%s:simd%d" % ("t%d" % w if w < 14 else "p%d" % (w - 14), (h >> 4) & 3))
        cu = "cu%d sh%d se%d xcc%d" % ((h >> 8) & 15, (h >> 12) & 1, (h >> 13) & 7, (hwx >> 32) & 15)
    print("== workgroup %d on %s:  %s" % (wg, cu, "  ".join(where)))
    end = {w: t[w, :ng] for w in range(W)}
    poll = {w: t[w, 256:256 + ng] for w in range(W)}
    waited = {w: t[w, 512:512 + ng] for w in range(W)}
    bstart = {w: t[w, 768:768 + nblk] for w in range(W)}
    t0 = min(int(end[w][end[w] > 0].min()) for w in range(W) if (end[w] > 0).any())
    for w in range(W):
        gs = np.nonzero(end[w] > 0)[0]
        if len(gs) < 8:
            continue
        # bodies: a start stamp, then the ends of its 2 nb groups, then the next body's start
        starts = [j for j in range(nblk) if bstart[w][j] > 0]
        by_nb = {}
        for a_, b_ in zip(starts, starts[1:] + [None]):
            nb_ = (b_ - a_) if b_ is not None else None
            if nb_ is None or nb_ not in (1, 2) or any(end[w][2 * a_ + q_] <= 0 for q_ in range(2 * nb_)):
                continue
            pts = [bstart[w][a_]] + [end[w][2 * a_ + q_] for q_ in range(2 * nb_)] + [bstart[w][b_]]
            by_nb.setdefault(nb_, []).append(np.diff(pts))
        pw = np.array([waited[w][g] - poll[w][g] for g in gs if waited[w][g] > 0 and poll[w][g] > 0])
        n_poll = int((poll[w][gs] > 0).sum())
        desc = []
        for nb_, rows_ in sorted(by_nb.items()):
            m_ = np.median(np.array(rows_), axis=0)
            desc.append("%d bodies of %d rows: start->g0 %4.0f | %s | last group end -> next start %4.0f  = %5.0f cycles (%4.1f per row)"
                        % (len(rows_), 32 * nb_, m_[0], " ".join("%4.0f" % x for x in m_[1:-1]), m_[-1], m_.sum(), m_.sum() / (32 * nb_)))
        print("  tile %d: groups %3d..%3d, ends +%7d .. +%7d | %s | polls %3d, waited at %3d: median %4.0f cycles, sum %6d"
              % (w, gs[0], gs[-1], end[w][gs[0]] - t0, end[w][gs[-1]] - t0, " ;; ".join(desc), n_poll, len(pw), med(pw), int(pw.sum()) if len(pw) else 0))
    for w in range(1, W):
        # slack at the poll: how long before my poll the neighbour finished the group I ask for (negative: I wait)
        both = np.nonzero((poll[w] > 0) & (end[w - 1] > 0))[0]
        if len(both) == 0:
            continue
        slack = (poll[w] - end[w - 1])[both]
        print("  tile %d polls tile %d: the neighbour's group was finished %5.0f cycles before the poll (median; p10 %5.0f, p90 %5.0f); late for %3d of %3d polls"
              % (w, w - 1, med(slack), np.percentile(slack, 10), np.percentile(slack, 90), int((slack < 0).sum()), len(slack)))
    for p in (14, 15):
        pub = t[p, :nblk]
        if not (pub > 0).any():
            continue
        sp = t[p, 120:120 + nblk]
        ok = (pub > 0) & (sp > 0)
        busy = (pub - sp)[ok]              # from "began to look for space" to "published": space wait + write
        idx = np.nonzero(ok)[0]
        gap = np.array([sp[j + 1] - pub[j] for j in idx if j + 1 < nblk and sp[j + 1] > 0])   # publish -> next wait_space: load issue
        lead = []
        for jb in idx:
            behind = [int(((end[w] > 0) & (end[w] <= pub[jb])).sum()) // 2 + (int(np.nonzero(end[w] > 0)[0][0]) // 2 if (end[w] > 0).any() else 0)
                      for w in range(W) if (end[w] > 0).any()]
            lead.append(jb + 1 - min(behind))
        print("  producer %d: block 0 published at +%d, the last at +%d; per block: look-for-space + write %5.0f (p90 %5.0f), then %4.0f to the next look;"
              " blocks ahead of the slowest tile when publishing: median %.0f, min %d, max %d"
              % (p - 14, pub[idx[0]] - t0, pub[idx[-1]] - t0, med(busy), np.percentile(busy, 90), med(gap), med(lead), min(lead), max(lead)))
    if tl_wg >= 0:
        print("  timeline (cycles since the first stamp): block | per tile: start, poll0 (+wait), end0, poll1 (+wait), end1")
        for j in range(nblk):
            row = []
            for w in range(W):
                if bstart[w][j] <= 0:
                    row.append("%38s" % "-")
                    continue
                f = lambda x: x - t0 if x > 0 else -1
                row.append("%7d %6d+%4d %7d %6d+%4d %7d" % (f(bstart[w][j]), f(poll[w][2 * j]) , max(0, waited[w][2 * j] - poll[w][2 * j]) if waited[w][2 * j] > 0 else 0,
                                                          f(end[w][2 * j]), f(poll[w][2 * j + 1]), max(0, waited[w][2 * j + 1] - poll[w][2 * j + 1]) if waited[w][2 * j + 1] > 0 else 0, f(end[w][2 * j + 1])))
            print("  %3d | %s" % (j, " | ".join(row)))
