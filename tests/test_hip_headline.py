"""GPU: the configuration that produces the headline number, checked in full.

BASELINE.json configs[2] exactly as ``bench.py`` runs it -- 512 DISTINCT segments x 3000 frames x vocab 32
(C = 640), checkpoint mode, ``ctcfa_plan_run_pipelined`` (the backtrack of step k beside the fill of step k+1 at
priority 0, four rotating workspaces, per-workspace error words), a different input set every step -- with EVERY
segment of several steps compared with the oracle (``oracle_c.get_segments``: what the reference reaches through
/root/reference/src/iterative_utterance_alignment.py:216).  And the self-spawning multi-rank launcher of
``bench.py --gpus N`` end to end, so that a scaling run does not fail on plumbing.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DUR = 320.4769 / 16000
SCORE_TOL = 1e-4   # north_star: frame indices bit-exact, confidence scores within 1e-4


def test_bench_self_spawned_two_ranks():
    """`python bench.py --gpus 2` from a bare shell: the GPU-free parent times the CPU baseline, starts two rank
    processes (here both on the one GPU, over gloo: CTCFA_BENCH_REHEARSAL), relays ONE JSON line.  Runs first in this
    file: the ranks are processes of their own, whatever this process has done to the GPU."""
    env = dict(os.environ, CTCFA_BENCH_REHEARSAL="1", CTCFA_BENCH_TIMEOUT="600")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1",
                          "--spinup-steps", "0", "--cpu-sample", "8", "--input-sets", "2", "--check-segments", "4"],
                         cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["ranks_seen"] == 2 and j["steps"] == 5
    assert j["cpu_baseline"] and "error" not in j["cpu_baseline"] and j["cpu_baseline"]["value"] > 0
    assert j["config"]["parity"] and j["config"]["gather"]
    assert j["roofline"]["frac"] > 0 and j["value"] > 0


def test_headline_schedule_full_oracle_check(pkg, oracle, engine):
    """Seven consecutive pipelined steps over four rotating input sets of 512 distinct segments each (the bench's own
    seeds), every step into an output set of its own, one flush at the end; then EVERY segment of steps 2, 5 and 6
    (input sets 2, 1, 2 -- workspaces 2, 1, 2 of the four) against the oracle."""
    import torch
    syn = pkg.synthetic
    B, T, V, U, n = 512, 3000, 32, 22, 28
    NIN, STEPS, CHECK = 4, 7, (2, 5, 6)
    host_sets = [syn.make_uniform_batch(B, T, V, U, n, seed0=64 * k * B) for k in range(NIN)]   # rank 0's sets in bench.py
    C = host_sets[0][1].shape[1]
    config = pkg.CtcSegmentationParameters(index_duration=DUR)
    plan = engine.plan(config.to_native(), V, [T] * B, [C] * B, [U] * B)
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    d_in = [(torch.from_numpy(h[0].reshape(-1)).to(dev), torch.from_numpy(h[1].astype(np.int32).reshape(-1)).to(dev),
             torch.from_numpy(h[2].astype(np.int32).reshape(-1)).to(dev)) for h in host_sets]
    outs = [dict(fol=torch.zeros(B * C, dtype=torch.int32, device=dev), cp=torch.zeros(B * T, dtype=torch.float32, device=dev),
                 st=torch.zeros(B * T, dtype=torch.int32, device=dev), seg=torch.zeros(3, B * U, dtype=torch.float64, device=dev),
                 te=torch.zeros(B, dtype=torch.int32, device=dev), status=torch.full((B,), -7, dtype=torch.int32, device=dev))
            for _ in range(STEPS)]
    for _ in range(3):   # (a few untimed rounds first: the steps checked below run beside a backtrack, as in the bench)
        for i in range(STEPS):
            d_lpz, d_lab, d_ub = d_in[i % NIN]
            o = outs[i]
            plan.run_device(d_lpz.data_ptr(), d_lab.data_ptr(), d_ub.data_ptr(), o["fol"].data_ptr(), o["cp"].data_ptr(),
                            o["st"].data_ptr(), o["seg"][0].data_ptr(), o["seg"][1].data_ptr(), o["seg"][2].data_ptr(),
                            o["te"].data_ptr(), o["status"].data_ptr(), stream, pipelined=True)
    plan.flush(stream)
    torch.cuda.synchronize()
    ocfg = oracle.make_config(index_duration=DUR)
    for i in range(STEPS):
        assert (outs[i]["status"].cpu().numpy() == 0).all(), f"step {i}: non-OK status"
    for i in CHECK:
        o = outs[i]
        h_lpz, h_gt, h_ub = host_sets[i % NIN]
        fol = o["fol"].cpu().numpy().reshape(B, C)
        cp = o["cp"].cpu().numpy().reshape(B, T)
        st = o["st"].cpu().numpy().reshape(B, T)
        seg = o["seg"].cpu().numpy().reshape(3, B, U)
        te = o["te"].cpu().numpy()
        for b in range(B):
            r = oracle.get_segments(h_lpz[b], h_gt[b], h_ub[b], ocfg)
            assert r["status"] == 0
            assert te[b] == r["t_end"], (i, b)
            assert np.array_equal(fol[b], r["frame_of_label"]), f"step {i} segment {b}: frame indices differ"
            assert np.array_equal(cp[b].astype(np.float64), r["char_probs"]), f"step {i} segment {b}: char_probs"
            assert np.array_equal(st[b], r["state"]), f"step {i} segment {b}: state list"
            assert np.array_equal(seg[0][b], r["seg_start"]) and np.array_equal(seg[1][b], r["seg_end"]), (i, b)
            np.testing.assert_allclose(seg[2][b], r["seg_score"], rtol=0, atol=SCORE_TOL)
    # the steps that share an input set must agree with each other bit for bit (sets 0..2 are used twice)
    for i in range(NIN, STEPS):
        for k in ("fol", "cp", "st", "seg", "te"):
            assert torch.equal(outs[i][k], outs[i - NIN][k]), (i, k)
    plan.close()
