#!/usr/bin/env python3
"""bench.py -- the reference's headline metric on MI355X (BASELINE.json).

metric : aligned audio hours/sec (CTC DP frames/s), whole job over all N GPUs
workload: --workload synthetic (default) = BASELINE.json configs[2] -- synthetic DP-only, 512 segments
          x 3000 frames x vocab 32 (C = 640 label columns, 22 utterances/segment), per GPU (weak scaling).
          The other configurations, DP-only on synthetic emissions (audio and the HF model are not
          available offline):
            replay  configs[1]: the recorded window sequence of the reference's 519 s sample file
                    (183 DP calls, tests/golden/replay_windows.json), --files copies in lockstep, every
                    round followed by the host read-back the anchor state machine needs;
            words   configs[3]: 10 000 word-level rows (T ~ U[100,750], 3 or 5 pieces), rows sharded
                    over the ranks (strong scaling);
            corpus  configs[4]: 100 h of windows drawn like the replay sequence, sharded over the ranks.
step   : one pass of the hot path (trellis fill + backtrack + utterance scoring) over that
          batch, inputs resident in HBM when the timed region starts.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

INDEX_DURATION = 320.4769 / 16000  # s per frame (wav2vec2: 215040 samples -> 671 frames)
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="synthetic", choices=["synthetic", "replay", "words", "corpus"])
    ap.add_argument("--files", type=int, default=64, help="replay: audio files advanced in lockstep per GPU")
    ap.add_argument("--copy-results", action="store_true",
                    help="replay: the scores go to HBM and come back with a copy per round (default: the backtrack kernel "
                         "writes them straight into pinned host memory, which is all the state machine reads)")
    ap.add_argument("--speculate", type=int, default=0,
                    help="replay: every DP request also carries the n texts the repeat loop may ask for next (the window "
                         "minus its last 1..n utterances) -- shared emissions and trellis fill, foreseen requests cost no round")
    ap.add_argument("--rows", type=int, default=10000, help="words: TSV rows of the whole job")
    ap.add_argument("--hours", type=float, default=100.0, help="corpus: audio hours of the whole job")
    ap.add_argument("--per-launch", type=int, default=0, help="words / corpus: segments per launch (0 = 2500 / 2048)")
    ap.add_argument("--segments", type=int, default=512, help="segments per GPU")
    ap.add_argument("--frames", type=int, default=3000)
    ap.add_argument("--vocab", type=int, default=32)
    ap.add_argument("--utts", type=int, default=22)
    ap.add_argument("--utt-len", type=int, default=28)
    ap.add_argument("--alphabet", type=int, default=0,
                    help="synthetic: the texts use the first N non-blank vocabulary entries only (a character model's vocabulary "
                         "holds more symbols than its texts show); texts of at most 31 entries beside the blank get a narrowed plan")
    ap.add_argument("--cols-per-lane", type=int, default=int(os.environ.get("CTCFA_K", "0")))
    ap.add_argument("--cpu-sample", type=int, default=512, help="segments timed on the CPU oracle (0 = skip)")
    ap.add_argument("--spinup-steps", type=int, default=1024,
                    help="untimed passes of the same step before the warm-up steps (~0.25 s), so that the timed "
                         "region sees the card's sustained clocks and not the ramp from idle (0 = off)")
    ap.add_argument("--no-gather", action="store_true", help="skip the RCCL gather of segment boundaries")
    ap.add_argument("--gather-every", type=int, default=64,
                    help="N > 1: segment boundaries of this many steps travel in one all-gather "
                         "(fewer, larger collectives; every step's results are still gathered inside the timed region)")
    ap.add_argument("--timing-stride", type=int, default=0,
                    help="HIP-event bracket every n-th launch of the timed region (0 = 4, or 1 for short runs)")
    ap.add_argument("--serial", action="store_true",
                    help="one stream: fill then backtrack per step (default: backtrack of step k overlaps fill of k+1)")
    ap.add_argument("--from-max-t", action="store_true",
                    help="backtrack_from_max_t: every path ends in the last frame (the longest backtrack); "
                         "the default recipe's paths end near frame 1950 of 3000")
    ap.add_argument("--input-sets", type=int, default=4,
                    help="synthetic: distinct emission / label sets in HBM, one per step in rotation (4 x 197 MB = 788 MB > the "
                         "256 MiB Infinity Cache: every step's emissions come from HBM, not from the MALL)")
    ap.add_argument("--check-segments", type=int, default=32,
                    help="synthetic: segments of EACH of the last two timed steps (two different input sets) compared with the oracle")
    ap.add_argument("--no-check", action="store_true",
                    help="kernel-tuning only: skip the status/parity gate (ablated builds give wrong results)")
    return ap.parse_args()


def issue_entry(kernel_ms_avg, rows, cols_per_lane, tiles_per_simd):
    """The kernel's OTHER ceiling ("VALU utilisation is reported beside it"): the fill is bound by
    instruction issue, not by HBM.  `valu` = SQ counters of the fill kernel reduced per launch (tools/collect_issue.sh ->
    profiles/r*_fill_issue.json); `issue_floor_ms` = trellis rows x the time one row of the SAME row loop takes in
    isolation (tools/row_rate: same ISA, `tiles_per_simd` waves per SIMD, nothing to wait for), `issue_frac` = that floor
    over the measured kernel time."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_fill_issue.json")))
    if not files:
        return {}
    try:
        j = json.load(open(files[-1]))
        rr = j["row_rate"]
        key = "K%d_w%d" % (cols_per_lane, max(1, min(4, tiles_per_simd)))
        ns_row = rr["ns_per_row"][key]
        floor_ms = rows * ns_row * 1e-6
        return {"valu": dict(j["valu"], source="profiles/" + os.path.basename(files[-1])),
                "issue_floor_ms": floor_ms, "issue_frac": floor_ms / kernel_ms_avg,
                "issue_floor_note": "%d rows x %.1f ns (%s cycles) per row of the isolated row loop at %d tile waves per SIMD (tools/row_rate, %s)"
                                    % (rows, ns_row, rr["cycles_per_row"].get(key), tiles_per_simd, key)}
    except Exception as exc:   # (a profile of another shape: no figure rather than a wrong one)
        return {"issue_note": "no issue-floor figure: " + repr(exc)}


def roofline_entry(fill_ms, bt_ms, ms_per_step, stride, fill_bytes, step_bytes, default_workload, input_sets=1,
                   rows=0, cols_per_lane=0, serial=False, tiles_per_simd=3, emission_bytes=0):
    """The `roofline` object.  `achieved` / `frac` belong to the DOMINANT KERNEL: the fill's own
    algorithmic bytes (emissions read once, 4TV, + the 1-bit-per-cell trace written once, TC/8) over the
    fill's own average launch duration.  `step_*` is the whole step: SURVEY section 8(d)'s per-segment figure
    (the backtrack's 4T + 8C + 4T included) over the step time.  `traffic` = HBM bytes per launch from
    the PMC counters (tools/collect_traffic.sh, separate --pmc passes, FETCH_SIZE x 2 on gfx950),
    fill and backtrack kernels, committed under profiles/ for the default workload."""
    import glob
    fill_s = float(np.mean(fill_ms)) * 1e-3
    achieved = fill_bytes / fill_s / 1e9
    step_achieved = step_bytes / (ms_per_step * 1e-3) / 1e9
    traffic, src, detail = None, None, None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if files and default_workload:
        try:
            j = json.load(open(files[-1]))
            fill_t = j["fill_kernel"]["hbm_bytes_per_launch"]
            bt_t = j.get("backtrack_kernel", {}).get("hbm_bytes_per_launch")
            traffic = fill_t
            detail = {"fill_kernel": fill_t, "backtrack_kernel": bt_t,
                      "fill_vs_fill_algorithmic": fill_t / fill_bytes,
                      "step_vs_step_algorithmic": ((fill_t + bt_t) / step_bytes) if bt_t else None,
                      "schedule": j.get("schedule")}
            src = "profiles/" + os.path.basename(files[-1]) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 FETCH x2)"
        except Exception:
            traffic = None
    issue = issue_entry(float(np.mean(fill_ms)), rows, cols_per_lane, tiles_per_simd) if (rows and default_workload) else {}
    resident = input_sets * (emission_bytes or fill_bytes)
    return {"bound": "hbm", "kernel": "ctcfa::fill_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_detail": detail, "traffic_source": src,
            "traffic_note": ("%d input sets in rotation: %.0f MB of emissions between two uses of the same bytes, %s the 256 MiB "
                             "Infinity Cache -- %s" % (input_sets, resident / 1e6, "beyond" if resident > 268e6 else "UNDER",
                                                       "the reads cross HBM" if resident > 268e6 else
                                                       "fabric bytes, MALL hits included")),
            **issue,
            "algorithmic_bytes_per_launch": fill_bytes,
            "algorithmic_bytes_note": "fill kernel only: 4TV + TC/8 per segment (the whole step, backtrack included: step_algorithmic_bytes)",
            "kernel_ms_avg": float(np.mean(fill_ms)), "kernel_ms_min": float(np.min(fill_ms)),
            "kernel_ms_samples": int(len(fill_ms)),
            "kernel_ms_sampling": f"HIP events around every {stride}-th launch of the timed region",
            "backtrack_kernel_ms_avg": float(np.mean(bt_ms)),
            "step_algorithmic_bytes": step_bytes, "step_achieved": step_achieved, "step_frac": step_achieved / HBM_PEAK_GBS}


def run_grouped(args, pkg, torch, rank, world, local_rank, rehearsal, cpu):
    """--workload replay | words | corpus: launches of ragged segments ("groups"); a step is one pass
    over all groups of this rank."""
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    syn = pkg.synthetic
    V = args.vocab
    calls = json.load(open(os.path.join(ROOT, "tests", "golden", "replay_windows.json")))["calls"]
    sequential = False
    shared = None          # per group: emission_of (replay with --speculate)
    frames_override = None
    if args.workload == "replay":
        # every round = one DP call of the recorded sequence for `files` files in lockstep (weak scaling:
        # `files` per GPU); the next round needs this round's scores on the host (anchor state machine)
        ok_calls = [c for c in calls if c["C"] <= c["T"]]
        sequential, scaling = True, "weak"
        if args.speculate <= 0:
            base = syn.make_windows_like(ok_calls, V, seed=rank)
            groups = [[b] * args.files for b in base]
            name = ("BASELINE.json configs[1]: replay of the reference's sample file (519 s, 157 rows -> %d DP calls, "
                    "T %d..%d), %d files in lockstep per GPU, DP-only on synthetic emissions"
                    % (len(groups), min(c["T"] for c in calls), max(c["T"] for c in calls), args.files))
        else:
            # windows = runs of calls over the same emissions, each the previous text minus its last utterance
            # (iterative_utterance_alignment.py:203 loop).  A request carries `speculate` foreseen texts; a
            # window of L calls then takes ceil(L / (speculate + 1)) rounds.  Foreseen texts are computed
            # whether or not the state machine will ask for them -- the driver cannot know.
            chains = []
            for c in ok_calls:
                if chains and chains[-1][-1]["T"] == c["T"] and chains[-1][-1]["utts"][:-1] == c["utts"]:
                    chains[-1].append(c)
                else:
                    chains.append([c])
            heads = []
            for ch in chains:
                heads += [ch[k] for k in range(0, len(ch), args.speculate + 1)]
            base = syn.make_windows_like(heads, V, seed=rank)
            groups, shared = [], []
            for lpz, gt, ub in base:
                U = len(ub) - 1
                members = [(lpz, gt[:ub[k] + 1].copy(), ub[:k + 1].copy())
                           for k in range(U, max(U - args.speculate, 0) - 1, -1) if k >= 1]
                groups.append(members * args.files)
                shared.append([f * len(members) for f in range(args.files) for _ in members])
            frames_override = sum(c["T"] for c in ok_calls) * args.files   # the recorded calls: every one is answered
            name = ("BASELINE.json configs[1]: replay of the reference's sample file (519 s, 157 rows -> %d DP calls in "
                    "%d windows), %d files in lockstep per GPU, every request with %d foreseen text(s) over the same "
                    "emissions and fill: %d rounds; frames counted = the recorded calls'; DP-only on synthetic emissions"
                    % (len(ok_calls), len(chains), args.files, args.speculate, len(groups)))
    elif args.workload == "words":
        rows = syn.make_word_rows(args.rows, V)
        costs = [s[0].shape[0] * len(s[1]) for s in rows]
        mine = pkg.sharding.assign_units(costs, world)[rank]
        segs = sorted((rows[i] for i in mine), key=lambda s: -s[0].shape[0] * len(s[1]))
        per = args.per_launch or 2500
        groups = [segs[k:k + per] for k in range(0, len(segs), per)]
        scaling = "strong"
        name = ("BASELINE.json configs[3]: word-level alignment of %d utterances (T ~ U[100,750], 3 or 5 pieces), "
                "rows cost-sharded over %d GPU(s), DP-only on synthetic emissions" % (args.rows, world))
    else:
        ok = [c for c in calls if c["C"] <= c["T"]]
        frames_rank = int(args.hours * 3600.0 / INDEX_DURATION / world)
        drawn = syn.draw_corpus_calls(ok, frames_rank, seed=rank)
        pool = syn.make_windows_like(drawn[:2048], V, seed=rank)      # 2048 distinct windows, tiled over the share
        segs = sorted((pool[i % len(pool)] for i in range(len(drawn))), key=lambda s: -s[0].shape[0] * len(s[1]))
        per = args.per_launch or 2048
        groups = [segs[k:k + per] for k in range(0, len(segs), per)]
        scaling = "strong"
        name = ("BASELINE.json configs[4]: %.0f h corpus as a window stream drawn like the sample file's "
                "(%d windows on this GPU), sharded over %d GPU(s), DP-only on synthetic emissions"
                % (args.hours, len(segs), world))

    cfg = pkg.CtcSegmentationParameters(index_duration=INDEX_DURATION)
    eng = pkg._native.Engine(local_rank)
    stream = torch.cuda.current_stream()
    G = []
    n_fills = 0
    for gi, g in enumerate(groups):
        Ts = [s[0].shape[0] for s in g]
        Cs = [len(s[1]) for s in g]
        Us = [len(s[2]) - 1 for s in g]
        em = shared[gi] if shared else None
        plan = eng.plan(cfg.to_native(), V, Ts, Cs, Us, force_cols_per_lane=args.cols_per_lane, emission_of=em,
                        labels=np.concatenate([s[1] for s in g]).astype(np.int32) if em else None)
        n_fills += plan.sharing()[0]
        d_lpz = torch.from_numpy(np.concatenate([s[0].reshape(-1) for b, s in enumerate(g) if not em or em[b] == b])).to(dev)
        d_lab = torch.from_numpy(np.concatenate([s[1] for s in g]).astype(np.int32)).to(dev)
        d_ub = torch.from_numpy(np.concatenate([s[2] for s in g]).astype(np.int32)).to(dev)
        nT, nC, nU, B = sum(Ts), sum(Cs), max(1, sum(Us)), len(g)
        direct = sequential and not args.copy_results   # scores written by the kernel into pinned host memory
        outs = [dict(fol=torch.empty(nC, dtype=torch.int32, device=dev), cp=torch.empty(nT, dtype=torch.float32, device=dev),
                     seg=(torch.zeros(3, nU, dtype=torch.float64).pin_memory() if direct
                          else torch.empty(3, nU, dtype=torch.float64, device=dev)),
                     te=torch.empty(B, dtype=torch.int32, device=dev),
                     status=torch.empty(B, dtype=torch.int32, device=dev)) for _ in range(3)]
        G.append(dict(plan=plan, lpz=d_lpz, lab=d_lab, ub=d_ub, outs=outs, frames=nT, segs=g, n=0,
                      host=torch.empty(3, nU, dtype=torch.float64).pin_memory() if sequential else None))
    frames_step = frames_override if frames_override is not None else sum(g["frames"] for g in G)
    pipelined = not args.serial and not sequential

    for g in G:   # (argument conversion once per output slot, not per launch)
        g["run"] = [g["plan"].bind(g["lpz"].data_ptr(), g["lab"].data_ptr(), g["ub"].data_ptr(), o["fol"].data_ptr(),
                                   o["cp"].data_ptr(), None, o["seg"][0].data_ptr(), o["seg"][1].data_ptr(),
                                   o["seg"][2].data_ptr(), o["te"].data_ptr(), o["status"].data_ptr(), stream.cuda_stream,
                                   pipelined=pipelined) for o in g["outs"]]

    def run_group(g):
        o = g["outs"][g["n"] % 3]
        g["run"][g["n"] % 3]()
        g["n"] += 1
        if sequential:   # the state machine reads this round's scores before it can form the next window
            if args.copy_results:
                g["host"].copy_(o["seg"], non_blocking=True)
            stream.synchronize()

    def step():
        for g in G:
            run_group(g)

    def drain():
        if pipelined:
            for g in G:
                g["plan"].flush(stream.cuda_stream)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    spin_budget = 0.3   # seconds of untimed load: clock ramp from idle (a fixed count per rank: no collectives inside)
    t0 = time.perf_counter()
    n_spin = 0
    while time.perf_counter() - t0 < spin_budget and n_spin < args.spinup_steps:
        step()
        torch.cuda.synchronize()
        n_spin += 1
    for _ in range(args.warmup):
        step()
    drain()
    barrier()
    # (events around every launch cost a short run ~4 %: every fourth launch from 17 steps on)
    stride = args.timing_stride if args.timing_stride > 0 else (1 if args.steps <= 8 else 2 if args.steps <= 16 else 4)
    n_timed = min((args.steps + stride - 1) // stride, 256)
    for g in G:
        g["plan"].set_timing(max(n_timed, 4))
        g["plan"].set_timing_stride(stride)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        ft = torch.tensor([float(frames_step)], dtype=torch.float64, device=dev)
        dist.all_reduce(ft)
        frames_all = float(ft.item())
    else:
        frames_all = float(frames_step)
    fill_ms = np.zeros(n_timed)
    bt_ms = np.zeros(n_timed)
    for g in G:   # per recorded step: the sum over this rank's launches
        f, b = g["plan"].get_timings(n_timed)
        fill_ms += f
        bt_ms += b

    # ---- parity gate on the timed inputs (a sample; the full-size checks are tests/test_hip_workloads.py)
    parity = None
    if rank == 0 and not args.no_check:
        from oracle import oracle_c
        ocfg = oracle_c.make_config(index_duration=INDEX_DURATION)
        checked = 0
        for g in (G[0], G[len(G) // 2], G[-1]):
            o = g["outs"][(g["n"] - 1) % 3]
            status = o["status"].cpu().numpy()
            assert (status == 0).all(), "non-OK status in the benchmark batch"
            fol = o["fol"].cpu().numpy()
            seg = o["seg"].cpu().numpy()
            co = np.concatenate([[0], np.cumsum([len(s[1]) for s in g["segs"]])])
            uo = np.concatenate([[0], np.cumsum([len(s[2]) - 1 for s in g["segs"]])])
            for b in (0, len(g["segs"]) - 1):
                lpz, gt, ub = g["segs"][b]
                ref = oracle_c.get_segments(lpz, gt, ub, ocfg)
                assert np.array_equal(fol[co[b]:co[b + 1]], ref["frame_of_label"]), "frame indices differ from oracle"
                assert np.allclose(seg[2][uo[b]:uo[b + 1]], ref["seg_score"], rtol=0, atol=1e-4)
                checked += 1
        parity = f"{checked} segments == oracle (frames bit-exact, scores<=1e-4)"

    if rank == 0:
        fps = frames_all * args.steps / dt
        fill_bytes = sum(4 * s[0].shape[0] * V + s[0].shape[0] * len(s[1]) // 8
                         for gi, g in enumerate(G) for b, s in enumerate(g["segs"]) if not shared or shared[gi][b] == b)
        step_bytes = sum(g["plan"].info["algorithmic_bytes"] for g in G)
        shapes = sorted({(g["plan"].info["cols_per_lane"], g["plan"].info["waves_per_seg"]) for g in G})
        out = {
            "metric": "aligned audio hours/sec (CTC DP frames/s)",
            "value": fps * INDEX_DURATION / 3600.0, "unit": "audio-hours/s", "frames_per_s": fps,
            "n_gpus": world, **dist_info(dist, rehearsal), "steps": args.steps, "warmup": args.warmup,
            "spinup": {"steps": n_spin, "note": "untimed passes (<= 0.3 s) before the warm-up steps: clock ramp from idle"},
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": name, "launches_per_step": len(G), "segments_per_step_this_gpu": sum(len(g["segs"]) for g in G),
                       "fills_per_step_this_gpu": n_fills,
                       "frames_per_step_this_gpu": frames_step, "vocab": V, "tile_shapes_K_W": shapes,
                       "parallelism": f"unit-sharded x{world}", "parity": parity,
                       "schedule": (("one launch at a time; the scores of every round are on the host before the next one starts (the "
                                     "anchor state machine needs them): " +
                                     ("copied back from HBM" if args.copy_results else "written by the backtrack kernel into pinned host memory"))
                                    if sequential else
                                    ("serial" if args.serial else "backtrack(k) overlaps fill(k+1) on a second stream"))},
            "roofline": roofline_entry(fill_ms, bt_ms, dt / args.steps * 1e3, stride, fill_bytes, step_bytes, False),
            "cpu_baseline": cpu,
        }
        out["roofline"]["kernel_ms_note"] = "per step: sum over the step's %d launches" % len(G)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline_child(args):
    """The cpu_baseline leg (oracle on the host cores) runs as a child process BEFORE this process
    touches the GPU: its all-cores part forks workers, which a HIP-initialised process must not do."""
    import subprocess
    cmd = [sys.executable, "-m", "oracle.cpu_baseline", "--workload", args.workload, "--segments", str(args.segments),
           "--frames", str(args.frames), "--vocab", str(args.vocab), "--utts", str(args.utts), "--utt-len", str(args.utt_len),
           "--sample", str(args.cpu_sample)]
    try:
        out = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        return json.loads(out.stdout.decode().strip().splitlines()[-1])
    except Exception as exc:
        return {"error": repr(exc)}


def spawn_ranks(args):
    """`python bench.py --gpus N` from a bare shell (no torch.distributed.run around it): this process
    never touches the GPU.  It times the CPU baseline first (the host cores are then free of rank
    processes), starts the N ranks as CHILDREN -- one process per GPU, the reference's n_process workers
    (/root/reference/align_utterances.sh:57,127-137) -- relays rank 0's JSON line and leaves with the
    children's worst return code."""
    import socket
    import subprocess
    import tempfile
    n = args.gpus
    env0 = dict(os.environ)
    env0.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this pool
    cpu_file = None
    if args.cpu_sample > 0:
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
        cpu = cpu_baseline_child(args)
        fd, cpu_file = tempfile.mkstemp(prefix="ctcfa_cpu_", suffix=".json")
        with os.fdopen(fd, "w") as f:
            json.dump(cpu, f)
        env0["CTCFA_BENCH_CPU_BASELINE"] = cpu_file
    # (one build, before N ranks race for the lock)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "__graft_entry__.py")], cwd=ROOT, env=env0)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    # rank 0's stdout goes to a file, not a pipe: nobody has to drain it while the ranks run (a library that prints more
    # than a pipe buffer would otherwise block rank 0 in write() and the others in the rendezvous)
    out0_file = tempfile.TemporaryFile(prefix="ctcfa_rank0_")
    for r in range(n):
        env = dict(env0, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CTCFA_BENCH_SPAWNED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], cwd=ROOT, env=env,
                                      stdout=out0_file if r == 0 else subprocess.DEVNULL))
    # (a rank that dies before the first barrier would leave the others waiting in the rendezvous for ever: the
    # launcher watches its own children and ends the rest -- exactly these processes -- when one fails, or when the
    # whole job overruns CTCFA_BENCH_TIMEOUT seconds)
    import time
    deadline = time.monotonic() + float(os.environ.get("CTCFA_BENCH_TIMEOUT", "1500"))
    timed_out = False
    while any(p.poll() is None for p in procs):
        timed_out = time.monotonic() > deadline
        if timed_out or any(p.poll() not in (None, 0) for p in procs):
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            break
        time.sleep(0.2)
    rcs = []
    for p in procs:
        try:
            rcs.append(p.wait(timeout=30))
        except subprocess.TimeoutExpired:
            p.kill()
            rcs.append(p.wait())
    out0_file.seek(0)
    out0 = out0_file.read().decode(errors="replace")
    out0_file.close()
    if timed_out:
        sys.stderr.write("bench.py: the ranks overran CTCFA_BENCH_TIMEOUT and were ended\n")
        rcs = [rc or 124 for rc in rcs]
    if cpu_file:
        os.unlink(cpu_file)
    # (ONE JSON line on stdout: what libraries of rank 0 may have printed there -- gloo announces its peers -- goes to stderr)
    for line in out0.splitlines():
        (sys.stdout if line.startswith("{") else sys.stderr).write(line + "\n")
    sys.stdout.flush()
    bad = [rc for rc in rcs if rc != 0]
    raise SystemExit(bad[0] if bad else 0)


def dist_info(dist, rehearsal):
    """What the collective layer itself reports (the driver checks the rank count against --gpus)."""
    if dist is None:
        return {"ranks_seen": 1, "backend": None}
    return {"ranks_seen": dist.get_world_size(), "backend": ("gloo (rehearsal on one GPU)" if rehearsal else "nccl (RCCL)")}


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return spawn_ranks(args)   # GPU-free launcher; the ranks come back through main() with WORLD_SIZE set
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    cpu = None
    if rank == 0 and args.cpu_sample > 0:
        pre = os.environ.get("CTCFA_BENCH_CPU_BASELINE")   # timed by the launcher above, before the ranks existed
        if pre and os.path.exists(pre):
            cpu = json.load(open(pre))
        else:
            # (build the oracle first: the child only needs the C restatement)
            import subprocess
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
            cpu = cpu_baseline_child(args)
    import torch
    import __graft_entry__ as ge

    pkg = ge.build()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # Rehearsal mode for a 1-GPU box (not used by the driver): every rank on cuda:0, gloo instead
    # of RCCL -- exercises the multi-rank control flow of this script without N devices.
    rehearsal = os.environ.get("CTCFA_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if args.workload != "synthetic":
        return run_grouped(args, pkg, torch, rank, world, local_rank, rehearsal, cpu)
    dev = torch.device("cuda", local_rank)
    dist = None
    # CTCFA_BENCH_FORCE_DIST=1 (not used by the driver): the collective path with a process group of ONE rank -- the
    # RCCL calls of the N > 1 schedule on a one-GPU box
    use_dist = world > 1 or os.environ.get("CTCFA_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # ---- synthetic workload (SURVEY §8(d) recipe), this rank's shard ----------------------
    syn = pkg.synthetic
    B, T, V, U, n = args.segments, args.frames, args.vocab, args.utts, args.utt_len
    NIN = max(1, args.input_sets)
    # input set k of rank r: seeds (r + 64 k) B ... -- set 0 is the recipe's own batch
    host_sets = [syn.make_uniform_batch(B, T, V, U, n, seed0=(rank + 64 * k) * B, alphabet=args.alphabet or None)
                 for k in range(NIN)]
    lpz, gt, ub = host_sets[0]
    C = gt.shape[1]
    cfg = pkg.CtcSegmentationParameters(index_duration=INDEX_DURATION)
    cfg.backtrack_from_max_t = bool(args.from_max_t)
    eng = pkg._native.Engine(local_rank)
    # (a vocabulary above 32 entries whose texts keep to 31 of them -- --vocab 38 --alphabet 28, a character model's windows --
    # gets a narrowed plan: CTCFA_FLAG_TEXTS_OF_31_LABELS, include/ctcfa.h)
    most = max(len(np.unique(g[g != cfg.blank])) - 1 for h in host_sets for g in h[1])   # (g[0] = -1 is not a label)
    promise = V > 32 and most <= 31
    # (texts of 32 .. 62 labels over more than 64 entries: the 64-entry ring, which a plan created with labels gets -- set 0's)
    ring64 = V > 64 and 31 < most <= 62
    plan = eng.plan(cfg.to_native(), V, [T] * B, [C] * B, [U] * B, force_cols_per_lane=args.cols_per_lane,
                    texts_of_31_labels=promise, labels=host_sets[0][1].astype(np.int32).reshape(-1) if ring64 else None)
    info = plan.info

    d_in = [(torch.from_numpy(h[0].reshape(-1)).to(dev), torch.from_numpy(h[1].astype(np.int32).reshape(-1)).to(dev),
             torch.from_numpy(h[2].astype(np.int32).reshape(-1)).to(dev)) for h in host_sets]
    def alloc_outputs():
        return dict(fol=torch.empty(B * C, dtype=torch.int32, device=dev),
                    cp=torch.empty(B * T, dtype=torch.float32, device=dev),
                    te=torch.empty(B, dtype=torch.int32, device=dev),
                    status=torch.empty(B, dtype=torch.int32, device=dev))

    # consecutive steps write different output sets: the pipelined schedule keeps two steps in
    # flight and the gather of step i-2 may still be reading while step i runs -> three sets
    outs = [alloc_outputs(), alloc_outputs(), alloc_outputs()]
    NSETS = len(outs)
    # Segment boundaries (start, end, score) -- what the job hands on -- go into group buffers:
    # G consecutive steps fill one buffer, which then travels in ONE all-gather.  Three buffers:
    # one being filled, one whose last steps are still in flight, one being gathered.
    G = max(1, args.gather_every)
    seg_ring = [torch.empty(G, 3, B * U, dtype=torch.float64, device=dev) for _ in range(3)]
    gathered = torch.empty(world, G, 3, B * U, dtype=torch.float64, device=dev) if use_dist else None
    stream = torch.cuda.current_stream()
    comm = torch.cuda.Stream(device=dev) if use_dist else None
    pipelined = not args.serial
    do_gather = use_dist and not args.no_gather
    n_calls = [0]
    n_gathered = [0]   # groups handed to the collective so far

    def seg_of(i):
        return seg_ring[(i // G) % 3][i % G]

    def gather(g):
        # the path's only exchange: gather the final segment boundaries (the role of
        # merge_aligned_files.py:17-25), on its own stream so it overlaps later steps
        buf = seg_ring[g % 3]
        if pipelined:
            plan.flush(comm.cuda_stream)   # the backtracks run on the plan's own stream: comm waits for every one still in flight
        comm.wait_stream(stream)           # ... and, either way, for what the caller's stream holds (fills; serial: everything)
        with torch.cuda.stream(comm):
            if rehearsal:   # gloo has no all_gather_into_tensor for device tensors
                parts = [torch.empty_like(buf) for _ in range(world)]
                dist.all_gather(parts, buf)
            else:
                dist.all_gather_into_tensor(gathered.view(-1), buf.view(-1))
        n_gathered[0] = g + 1

    def step():
        i = n_calls[0]
        n_calls[0] += 1
        o = outs[i % NSETS]
        sg = seg_of(i)
        if do_gather and i % G == 0 and i >= 3 * G:
            stream.wait_stream(comm)   # the gather that last read this group buffer (group i/G - 3) has finished
        d_lpz, d_lab, d_ub = d_in[i % NIN]   # a different input set every step: the emissions come from HBM
        plan.run_device(d_lpz.data_ptr(), d_lab.data_ptr(), d_ub.data_ptr(), o["fol"].data_ptr(),
                        o["cp"].data_ptr(), None, sg[0].data_ptr(), sg[1].data_ptr(),
                        sg[2].data_ptr(), o["te"].data_ptr(), o["status"].data_ptr(),
                        stream.cuda_stream, pipelined=pipelined)
        if do_gather:
            if i % G == G - 1:
                gather(i // G)

    def drain():
        if do_gather:   # (first: the gathers order themselves behind the backtracks still in flight)
            for g in range(n_gathered[0], (n_calls[0] + G - 1) // G):   # groups not yet on their way (the last may be partial)
                gather(g)
        if pipelined:
            plan.flush(stream.cuda_stream)
        if comm is not None:
            stream.wait_stream(comm)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # The card idles at a few hundred MHz and takes ~50-100 ms of load to reach its sustained
    # clocks; a K-step region is only K x 0.23 ms long, so without this a short run times the ramp
    # (30 steps from idle: 0.288 ms/step; the same 30 steps after the spin-up: 0.233).
    # A fixed step count (not a wall-clock budget): every rank must issue the same collectives.
    for i in range(args.spinup_steps):
        step()
        if i % 64 == 63:
            torch.cuda.synchronize()
    # Kernel durations come from HIP events recorded around every `stride`-th launch of the timed
    # region: an event record is a packet the queue retires between two kernels (~5 us each), and
    # bracketing every launch would itself take ~4 % off the number being measured.
    # (events around every launch cost a short run ~4 %: every fourth launch from 17 steps on)
    stride = args.timing_stride if args.timing_stride > 0 else (1 if args.steps <= 8 else 2 if args.steps <= 16 else 4)
    n_timed = min((args.steps + stride - 1) // stride, 1024)
    # (the events exist before the warm-up and are recorded in it: their first use is the runtime's business -- a fresh
    # process paid up to 1.8 ms for it inside a 20-step region)
    plan.set_timing(max(n_timed, 4))
    plan.set_timing_stride(stride)
    for _ in range(max(args.warmup, 2 * stride)):
        step()
    drain()
    barrier()
    plan.set_timing(max(n_timed, 4))   # (the same events, from the first slot)
    plan.set_timing_stride(stride)
    n_calls[0] = 0
    n_gathered[0] = 0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    fill_ms, bt_ms = plan.get_timings(n_timed)
    # per-step samples from the events themselves: fill start of one recorded run to fill start of the next, / stride.
    # The wall-clock figure of a short region also pays the one backtrack nothing overlaps (the last); the median of
    # these does not, and a 20-step run gives the same number as a 1000-step one.
    step_ev = plan.get_step_intervals(n_timed) / stride if n_timed >= 2 else np.zeros(0)

    # ---- parity gate on the timed inputs: `--check-segments` segments of EACH of the last two timed steps (two
    # input sets, two output sets, two workspaces) against the oracle; every segment's status.  The full-size check of
    # the schedule (every segment of several steps) is tests/test_hip_parity.py::test_headline_schedule_full_oracle_check.
    parity = None
    n_checked = 0
    for back in (1, 2):
        i = args.steps - back
        if i < 0:
            continue
        o = outs[i % NSETS]
        status = o["status"].cpu().numpy()
        assert args.no_check or (status == 0).all(), "non-OK status in the benchmark batch"
        if rank == 0 and not args.no_check and args.check_segments > 0:
            from oracle import oracle_c
            ocfg = oracle_c.make_config(index_duration=INDEX_DURATION, backtrack_from_max_t=int(args.from_max_t))
            h_lpz, h_gt, h_ub = host_sets[i % NIN]
            fol = o["fol"].cpu().numpy().reshape(B, C)
            seg = seg_of(i).cpu().numpy().reshape(3, B, U)
            cp = o["cp"].cpu().numpy().reshape(B, T)
            te = o["te"].cpu().numpy()
            for b in sorted(set(np.linspace(0, B - 1, min(B, args.check_segments)).astype(int).tolist())):
                r = oracle_c.get_segments(h_lpz[b], h_gt[b], h_ub[b], ocfg)
                assert te[b] == r["t_end"], f"step {i} segment {b}: end frame differs from oracle"
                assert np.array_equal(fol[b], r["frame_of_label"]), f"step {i} segment {b}: frame indices differ from oracle"
                assert np.array_equal(cp[b].astype(np.float64), r["char_probs"]), f"step {i} segment {b}: char_probs differ"
                assert np.array_equal(seg[0][b], r["seg_start"]) and np.array_equal(seg[1][b], r["seg_end"])
                assert np.allclose(seg[2][b], r["seg_score"], rtol=0, atol=1e-4)
                n_checked += 1
    if n_checked:
        parity = f"{n_checked} segments of the last two timed steps == oracle (frames, char_probs bit-exact; scores<=1e-4)"

    if rank == 0:
        frames_total = world * B * T * args.steps
        fps = frames_total / dt
        value = fps * INDEX_DURATION / 3600.0
        out = {
            "metric": "aligned audio hours/sec (CTC DP frames/s)",
            "value": value, "unit": "audio-hours/s", "frames_per_s": fps,
            "n_gpus": world, **dist_info(dist, rehearsal), "steps": args.steps, "warmup": args.warmup,
            "spinup": {"steps": args.spinup_steps, "note": "untimed, before the warm-up steps: clock ramp from idle"},
            "ms_per_step": dt / args.steps * 1e3,
            "ms_per_step_events": ({"median": float(np.median(step_ev)), "mean": float(np.mean(step_ev)),
                                    "p10": float(np.percentile(step_ev, 10)), "p90": float(np.percentile(step_ev, 90)),
                                    "samples": int(len(step_ev)),
                                    "note": f"fill start to fill start of consecutive HIP-event-bracketed launches / {stride}"}
                                   if len(step_ev) else None),
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[2]: synthetic DP-only, %d segments x %d frames x "
                                   "vocab %d per GPU, C=%d label columns, %d utterances/segment"
                                   % (B, T, V, C, U),
                       "segments_per_gpu": B, "frames": T, "vocab": V, "label_columns": C,
                       "input_sets": NIN, "input_bytes_resident": int(NIN * B * T * V * 4),
                       "cols_per_lane": info["cols_per_lane"], "waves_per_segment": info["waves_per_seg"],
                       "narrowed_plan": bool((V > 32 and info["vocab_pitch"] == 34) or (V > 64 and info["vocab_pitch"] == 66)),
                       "parallelism": f"segment-sharded x{world}", "parity": parity,
                       "gather": (f"RCCL all-gather of (start, end, score), one per {G} steps" if do_gather else None),
                       "schedule": "serial" if args.serial else "backtrack(k) overlaps fill(k+1) on a second stream"},
            "roofline": roofline_entry(fill_ms, bt_ms, dt / args.steps * 1e3, stride,
                                       fill_bytes=B * (4 * T * V + T * C // 8), step_bytes=info["algorithmic_bytes"],
                                       default_workload=(B, T, V, U, n) == (512, 3000, 32, 22, 28), input_sets=NIN,
                                       rows=T - 1, cols_per_lane=info["cols_per_lane"], serial=args.serial,
                                       tiles_per_simd=-(-min(2, -(-B // 256)) * info["waves_per_seg"] // 4),
                                       emission_bytes=B * T * V * 4),
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
