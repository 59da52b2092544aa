"""TEST INFRASTRUCTURE: checks that do NOT go through the restatement's decision rule.

The oracle (oracle/) restates ctc-segmentation 1.7.1 from recollection -- "parity unpinned".  The
functions here give the DP results support that is independent of that restatement's backtrack:

* ``brute_force_optimum``: the value the recurrence of SURVEY Appendix A.2 defines,
  table[t, c] = max over all monotone paths (0,0) -> (t,c) of the summed step costs, obtained by
  ENUMERATING the paths (no dynamic programme, no residual rule), on emissions that are
  multiples of 1/4 so that every fp32 sum is exact;
* ``path_from_result``: the path a result (frame_of_label, t_end) describes, and its exact cost;
* ``sequential_fp32_sum``: the sum of char_probs along the returned path, accumulated in fp32 in
  frame order -- what the fill computed for table[t_end, C-1] if the path is the one it took.
* ``utterance_segments``: SURVEY Appendix A.4 (determine_utterance_segments) written out with NumPy's own
  ``mean`` -- the function the package calls -- from a result's frame_of_label / char_prob: the boundaries and
  scores of a result are checked without going through oracle/ at all (its pairwise-summation restatement
  included).
"""
import itertools

import numpy as np

NEG = -1e9


def utterance_segments(frame_of_label, char_prob, utt_begin, index_duration, n=30):
    """-> (start[U], end[U], score[U]) in float64, straight from Appendix A.4: timings = frame x index_duration;
    boundaries between the neighbouring labels' timings, clamped to +-0.5 s; score = the minimum over the
    sliding means of ``n`` frames of char_probs (np.mean), or the plain mean of a shorter span."""
    timings = np.asarray(frame_of_label, np.int64).astype(np.float64) * float(index_duration)
    cp = np.asarray(char_prob, np.float32).astype(np.float64)

    def compute_time(i, kind):
        middle = (timings[i] + timings[i - 1]) / 2
        return max(timings[i + 1] - 0.5, middle) if kind == "begin" else min(timings[i - 1] + 0.5, middle)

    starts, ends, scores = [], [], []
    ub = [int(u) for u in utt_begin]
    for u in range(len(ub) - 1):
        start, end = compute_time(ub[u], "begin"), compute_time(ub[u + 1], "end")
        start_t, end_t = int(round(start / index_duration)), int(round(end / index_duration))
        if end_t <= start_t:
            score = -10000000000.0
        elif end_t - start_t <= n:
            score = float(np.mean(cp[start_t:end_t]))
        else:
            score = 0.0
            for t in range(start_t, end_t - n):
                score = min(score, float(np.mean(cp[t:t + n])))
        starts.append(start)
        ends.append(end)
        scores.append(score)
    return np.asarray(starts), np.asarray(ends), np.asarray(scores)


def step_costs(lpz, gt, blank=0, preamble=True):
    """stay[t, c], switch[t, c] (cost of arriving in column c at frame t) as Python floats."""
    T, C = lpz.shape[0], len(gt)
    stay = [[0.0] * C for _ in range(T)]
    sw = [[0.0] * C for _ in range(T)]
    for t in range(T):
        lb = float(lpz[t, blank])
        for c in range(C):
            if c == 0:
                stay[t][c] = 0.0 if preamble else max(lb, NEG)
                sw[t][c] = None          # nothing switches into the start column
            else:
                e = float(lpz[t, gt[c]])
                stay[t][c] = max(lb, e, NEG)
                sw[t][c] = e
    return stay, sw


def brute_force_optimum(lpz, gt, blank=0, preamble=True):
    """-> (best[t] for t in 0..T-1: the maximum over ALL monotone paths (0,0)->(t,C-1), or None when
    no path reaches the last column by frame t; t_end = first frame attaining max_t best[t])."""
    T, C = lpz.shape[0], len(gt)
    stay, sw = step_costs(lpz, gt, blank, preamble)
    best = [None] * T
    # a path = the frames at which it switches into columns 1..C-1: an increasing (C-1)-tuple of frames >= 1
    for t_last in range(1, T):
        tot_best = None
        for switches in itertools.combinations(range(1, t_last + 1), C - 1):
            tot, c, k = 0.0, 0, 0
            for t in range(1, t_last + 1):
                if k < C - 1 and switches[k] == t:
                    c += 1
                    k += 1
                    tot += sw[t][c]
                else:
                    tot += stay[t][c]
            if tot_best is None or tot > tot_best:
                tot_best = tot
        best[t_last] = tot_best
    reach = [b for b in best if b is not None]
    if not reach:
        return best, None
    top = max(reach)
    return best, next(t for t, b in enumerate(best) if b is not None and b == top)


def path_from_result(lpz, gt, frame_of_label, t_end, blank=0, preamble=True):
    """Exact cost of the path the result describes (frame_of_label[c] = frame of the switch into c)."""
    T, C = lpz.shape[0], len(gt)
    stay, sw = step_costs(lpz, gt, blank, preamble)
    fol = [int(f) for f in frame_of_label]
    assert fol[0] == 0
    assert all(fol[c] < fol[c + 1] for c in range(1, C - 1)), "switch frames must increase"
    assert C == 1 or (1 <= fol[1] and fol[C - 1] <= t_end), "path must end in the last column by t_end"
    tot, c = 0.0, 0
    for t in range(1, t_end + 1):
        if c + 1 < C and fol[c + 1] == t:
            c += 1
            tot += sw[t][c]
        else:
            tot += stay[t][c]
    assert c == C - 1
    return tot


def sequential_fp32_sum(char_probs, t_end, first=1):
    """(``first``: frame of the switch into column 1 -- with preamble_transition_cost_zero the table
    charges nothing before it, while char_probs report the blank posterior of those frames)"""
    acc = np.float32(0.0)
    for t in range(int(first), int(t_end) + 1):
        acc = np.float32(acc + np.float32(char_probs[t]))
    return float(acc)


def grid_case(rng, T, C, V, blank=0):
    """Emissions on a 1/4 grid in [-8, 0] (every partial sum is exact in fp32 and in float64),
    labels in [0, V) with the package's layout quirks allowed (repeated labels, blanks inside)."""
    lpz = (-rng.integers(0, 33, size=(T, V)) / 4.0).astype(np.float32)
    gt = np.concatenate([[-1], rng.integers(0, V, size=C - 1)]).astype(np.int64)
    return lpz, gt
