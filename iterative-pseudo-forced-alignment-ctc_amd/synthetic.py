"""Synthetic CTC-segmentation workloads (BASELINE.json configs[2]).

Audio and the HF acoustic model of the reference are unavailable offline
(/root/reference/data/sample/README.md:8-12; align_utterances.sh:74), so the DP is
exercised on seeded synthetic emission tensors of the reference's shapes:
``lpz = log_softmax(3*N(0,1) + 6*onehot(planted monotone path))`` in fp32 and label
columns laid out the way ``prepare_token_list`` lays them out
(``[-1]``, then ``blank`` + ids per utterance, then a closing ``blank``).
"""
import numpy as np


def make_labels(rng, n_utts, utt_len, vocab, blank=0, alphabet=None):
    """-> (gt int64 [C], utt_begin int64 [U+1]) with C = 1 + U*(1+n) + 1.  ``alphabet``: the texts use the first
    ``alphabet`` non-blank entries only (a character model's vocabulary holds more symbols than its texts show)."""
    gt = [-1]
    utt_begin = []
    for _ in range(n_utts):
        if gt[-1] != blank:
            gt.append(blank)
        utt_begin.append(len(gt) - 1)
        ids = rng.integers(1, vocab if alphabet is None else min(vocab, alphabet + 1), size=utt_len)
        if blank != 0:
            ids = np.where(ids == blank, 0, ids)
        gt.extend(int(i) for i in ids)
    if gt[-1] != blank:
        gt.append(blank)
    utt_begin.append(len(gt) - 1)
    return np.asarray(gt, np.int64), np.asarray(utt_begin, np.int64)


def make_emissions(rng, n_frames, vocab, gt, blank=0, noise=3.0, peak=6.0):
    """fp32 [T, V] log-posteriors with a planted monotone path through ``gt``."""
    T, C = int(n_frames), len(gt)
    logits = (noise * rng.standard_normal((T, vocab))).astype(np.float32)
    if C <= T and C > 1:
        # first frame of every column c >= 1, strictly increasing, frame 0 stays in column 0
        firsts = np.sort(rng.choice(np.arange(1, T), size=C - 1, replace=False))
        col = np.zeros(T, np.int64)
        col[firsts] = 1
        col = np.cumsum(col)
        sym = np.where(col > 0, gt[np.maximum(col, 1)], blank)
        is_first = np.zeros(T, bool)
        is_first[firsts] = True
        coin = rng.random(T) < 0.5
        hot = np.where(is_first | coin, sym, blank)
        logits[np.arange(T), hot] += np.float32(peak)
    m = logits.max(axis=1, keepdims=True)
    z = logits - m
    lse = np.log(np.exp(z.astype(np.float64)).sum(axis=1, keepdims=True)).astype(np.float32)
    return (z - lse).astype(np.float32)


def make_segment(seed, n_frames, vocab, n_utts, utt_len, blank=0, alphabet=None):
    """One (lpz, gt, utt_begin) triple; seeds follow SURVEY §8(d): 1234+b / 4321+b."""
    gt, utt_begin = make_labels(np.random.default_rng(4321 + seed), n_utts, utt_len, vocab, blank, alphabet)
    lpz = make_emissions(np.random.default_rng(1234 + seed), n_frames, vocab, gt, blank)
    return lpz, gt, utt_begin


def make_uniform_batch(batch, n_frames, vocab, n_utts, utt_len, seed0=0, blank=0, alphabet=None):
    """-> lpz [B,T,V] f32, gt [B,C] i64, utt_begin [B,U+1] i64."""
    segs = [make_segment(seed0 + b, n_frames, vocab, n_utts, utt_len, blank, alphabet) for b in range(batch)]
    return (np.stack([s[0] for s in segs]), np.stack([s[1] for s in segs]),
            np.stack([s[2] for s in segs]))


# BASELINE.json configs[2]: 512 segments x 3000 frames x vocab 32, C = 640 label columns
CONFIG3 = dict(batch=512, n_frames=3000, vocab=32, n_utts=22, utt_len=28)


def make_labels_from_lengths(rng, lengths, vocab, blank=0):
    """Label sequence for utterances of the given token counts (zero-length utterances allowed, as
    ``prepare_token_list`` produces them for items that tokenise to nothing)."""
    gt = [-1]
    utt_begin = []
    for n in lengths:
        if gt[-1] != blank:
            gt.append(blank)
        utt_begin.append(len(gt) - 1)
        ids = rng.integers(1, vocab, size=int(n))
        if blank != 0:
            ids = np.where(ids == blank, 0, ids)
        gt.extend(int(i) for i in ids)
    if gt[-1] != blank:
        gt.append(blank)
    utt_begin.append(len(gt) - 1)
    return np.asarray(gt, np.int64), np.asarray(utt_begin, np.int64)


def make_word_rows(n_rows, vocab=32, seed=0, blank=0):
    """BASELINE.json configs[3] (align_words.sh over 10 000 utterances): one segment per TSV row,
    T ~ U[100, 750] frames (2..15 s clips), the sentence cut into 3 or 5 pieces around the wanted word
    with one-token separators between them (word_level_alignment.py:69-84) -- SURVEY §8(d)."""
    rng = np.random.default_rng(9100 + seed)
    segs = []
    for _ in range(n_rows):
        T = int(rng.integers(100, 751))
        pieces = int(rng.choice([3, 5]))
        budget = max(pieces * 2, min(T // 4, 70))            # label columns stay well below the frame count
        lens = []
        for p in range(pieces):
            lens.append(int(rng.integers(2, max(3, budget // pieces))))
            lens.append(1)                                     # the "·" separator item
        gt, ub = make_labels_from_lengths(rng, lens, vocab, blank)
        segs.append((make_emissions(rng, T, vocab, gt, blank), gt, ub))
    return segs


def make_windows_like(calls, vocab=32, seed=0, blank=0):
    """Segments with the shapes of recorded DP calls (tests/golden/replay_windows.json entries:
    ``T`` frames, ``utts`` = tokens per utterance): the anchor iteration's window stream, DP-only."""
    rng = np.random.default_rng(9200 + seed)
    segs = []
    for c in calls:
        gt, ub = make_labels_from_lengths(rng, c["utts"], vocab, blank)
        segs.append((make_emissions(rng, int(c["T"]), vocab, gt, blank), gt, ub))
    return segs


def draw_corpus_calls(calls, frames_wanted, seed=0):
    """BASELINE.json configs[4] (100 h corpus): windows drawn with replacement from a recorded window
    sequence until they hold ``frames_wanted`` frames."""
    rng = np.random.default_rng(9300 + seed)
    out, total = [], 0
    while total < frames_wanted:
        c = calls[int(rng.integers(0, len(calls)))]
        out.append(c)
        total += int(c["T"])
    return out
