/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.
 *
 * CPU restatement (plain C99) of the CTC-segmentation dynamic program that the
 * reference delegates to the un-vendored PyPI package `ctc-segmentation==1.7.1`
 * (pin: /root/reference/requirements.txt:13) through
 * `speechbrain==0.5.11` (requirements.txt:87).  The package is not installed in
 * the build container and cannot be fetched (no network), and the reference's
 * own tests (src/test/test_ctc_segmentation.py:40-43) print results without
 * asserting on them, so there is NO golden vector from the reference for this
 * path: parity of this oracle against the real package is UNPINNED.  It follows
 * the package's published algorithm (functions named below) and is anchored on
 * the reference's call sites:
 *   src/iterative_utterance_alignment.py:201-219  (get_lpz / prepare_segmentation_task /
 *                                                   get_segments / task.set / str(task))
 *   src/word_level_alignment.py:89-103, src/search_on_speech.py:74-88
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link, load or call this file.  The product path (the HIP library under
 * iterative-pseudo-forced-alignment-ctc_amd/csrc) never does.
 *
 * Restated functions (ctc-segmentation 1.7.1):
 *   cython_fill_table            (ctc_segmentation_dyn.pyx)   -> oracle_fill_table
 *   ctc_segmentation             (ctc_segmentation.py)        -> oracle_ctc_segmentation
 *   determine_utterance_segments (ctc_segmentation.py)        -> oracle_determine_utterance_segments
 *
 * Arithmetic contract reproduced here:
 *   - the trellis is fp32 (np.float32 table, C `float` locals in the Cython loop);
 *   - the Cython sentinel is -1e9, the NumPy-side `max_prob` is -1e10;
 *   - the backtrack infers the transition taken from fp32 residuals
 *     |lpz - (table[t,c] - table[prev])|, strict '>' so ties go to STAY;
 *   - timings / char_probs are fp64 arrays; np.mean is pairwise fp64.
 *   - NumPy negative indices wrap; an out-of-range index is the package's
 *     IndexError, which doubles the window and finally re-raises.
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off, no fast-math).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_OK 0
#define ORACLE_AUDIO_SHORTER_THAN_TEXT 1 /* AssertionError("Audio is shorter than text!") */
#define ORACLE_INDEX_ERROR 2             /* IndexError re-raised at max_window_size          */

typedef struct {
    double max_prob;               /* -1e10 */
    double skip_prob;              /* -1e10 */
    int32_t min_window_size;       /* 8000  */
    int32_t max_window_size;       /* 100000 */
    double index_duration;         /* seconds per lpz frame */
    int32_t score_min_mean_over_L; /* 30 */
    int32_t blank;                 /* 0 */
    int32_t blank_transition_cost_zero;    /* flags bit 0 */
    int32_t preamble_transition_cost_zero; /* flags bit 1 */
    int32_t backtrack_from_max_t;
} oracle_config;

void oracle_default_config(oracle_config* c) {
    c->max_prob = -10000000000.0;
    c->skip_prob = -10000000000.0;
    c->min_window_size = 8000;
    c->max_window_size = 100000;
    c->index_duration = 0.025;
    c->score_min_mean_over_L = 30;
    c->blank = 0;
    c->blank_transition_cost_zero = 0;
    c->preamble_transition_cost_zero = 1;
    c->backtrack_from_max_t = 0;
}

/* Cython `max(a, b)` on C floats: the later operand wins only if greater. */
static inline float fmax_cy(float a, float b) { return (b > a) ? b : a; }

/*
 * cython_fill_table(table, lpz, ground_truth, offsets, blank, flags).
 * table: [W, C] fp32 row-major, pre-filled by the caller with max_prob.
 * lpz:   [T, V] fp32.  gt: [C, S] int64 (-1 padded).  offsets: [C] int64 out.
 * Returns t of the first maximum in the last column (c = C-1).
 * Negative indices wrap exactly as in Cython's default wraparound=True mode.
 */
int64_t oracle_fill_table(float* table, int64_t W, int64_t C, const float* lpz, int64_t T,
                          int64_t V, const int64_t* gt, int64_t S, int64_t* offsets,
                          int32_t blank, int32_t flags) {
    const float prob_max = -1000000000.0f;
    int64_t offset = 0, offset_sum = 0, higher_offset, lastArgMax = -1;
    float lastMax = 0.0f;
    const int blank_cost_zero = flags & 1;
    const int preamble_cost_zero = flags & 2;
    int64_t* cur_offset = (int64_t*)malloc(sizeof(int64_t) * (size_t)S);
    for (int64_t s = 0; s < S; ++s) cur_offset[s] = -1;
    /* `(lpz.shape[0] - table.shape[0]) / float(table.shape[1])` is a Python-float
     * (fp64) division assigned to a C float. */
    float mean_offset = (float)((double)(T - W) / (double)C);
    higher_offset = (int64_t)mean_offset + 1;
    table[0] = 0.0f;
    for (int64_t c = 0; c < C; ++c) {
        if (c > 0) {
            int64_t a = lastArgMax - W / 2;
            if (a < 0) a = 0;
            int64_t b = (T - W) - offset_sum;
            if (higher_offset < b) b = higher_offset;
            offset = (a < b) ? a : b;
            for (int64_t s = 0; s + 1 < S; ++s) cur_offset[s + 1] = cur_offset[s] + offset;
            cur_offset[0] = offset;
            offset_sum += offset;
        }
        offsets[c] = offset_sum;
        lastArgMax = -1;
        lastMax = 0.0f;
        for (int64_t t = (c == 0 ? 1 : 0); t < W; ++t) {
            float switch_prob = prob_max;
            float max_lpz_prob = prob_max;
            for (int64_t s = 0; s < S; ++s) {
                int64_t g = gt[c * S + s];
                if (g == -1) continue;
                float e = lpz[(t + offset_sum) * V + g];
                float p;
                if (t >= W - (cur_offset[s] - 1) || t - 1 + cur_offset[s] < 0) {
                    p = prob_max;
                } else {
                    int64_t pc = c - (s + 1);
                    if (pc < 0) pc += C; /* Cython wraparound */
                    p = table[(t - 1 + cur_offset[s]) * C + pc] + e;
                }
                switch_prob = fmax_cy(switch_prob, p);
                max_lpz_prob = fmax_cy(max_lpz_prob, e);
            }
            float stay_prob;
            if (t - 1 < 0) {
                stay_prob = prob_max;
            } else if (c == 0 && preamble_cost_zero) {
                stay_prob = 0.0f;
            } else {
                float stay_step = fmax_cy(lpz[(t + offset_sum) * V + blank], max_lpz_prob);
                /* flags bit 0 (gratis blank): a column labelled blank stays for free.
                 * [SURVEY A.6 U3: semantics recalled, unused by the reference's
                 *  production scripts (flags == 2 there).] */
                if (blank_cost_zero && S >= 1 && gt[c * S] == blank) stay_step = 0.0f;
                stay_prob = table[(t - 1) * C + c] + stay_step;
            }
            float v = fmax_cy(switch_prob, stay_prob);
            table[t * C + c] = v;
            if (lastArgMax == -1 || lastMax < v) {
                lastMax = v;
                lastArgMax = t;
            }
        }
    }
    free(cur_offset);
    return lastArgMax;
}

/* NumPy index semantics: wrap negatives once, flag out-of-range. */
static inline int64_t np_idx(int64_t i, int64_t n, int* err) {
    if (i < -n || i >= n) {
        *err = 1;
        return 0;
    }
    return i < 0 ? i + n : i;
}

/* Python `max(a, b)`: returns a unless b > a. */
static inline double pymax(double a, double b) { return (b > a) ? b : a; }

/*
 * ctc_segmentation(config, lpz, ground_truth): fill + backtrack with the
 * window-doubling retry.  Outputs (caller allocated):
 *   timings   [C] fp64 seconds      char_probs [T] fp64
 *   state     [T] int32: label id of the frame's switch, -1 for the self
 *             transition ("ε"), -2 where state_list keeps "" (untouched)
 *   frame_of_label [C] int32: (offsets[c] + t) of the switch that set timings[c], 0 if unset
 *   t_end_out: row of the backtrack start in the final (successful) attempt
 * Returns ORACLE_OK / ORACLE_AUDIO_SHORTER_THAN_TEXT / ORACLE_INDEX_ERROR.
 */
int oracle_ctc_segmentation(const oracle_config* cfg, const float* lpz, int64_t T, int64_t V,
                            const int64_t* gt, int64_t C, int64_t S, double* timings,
                            double* char_probs, int32_t* state, int32_t* frame_of_label,
                            int64_t* t_end_out) {
    const int32_t blank = cfg->blank;
    const int32_t flags =
        (cfg->blank_transition_cost_zero ? 1 : 0) + (cfg->preamble_transition_cost_zero ? 2 : 0);
    if (C > T && cfg->skip_prob <= cfg->max_prob) return ORACLE_AUDIO_SHORTER_THAN_TEXT;
    int64_t window_size = cfg->min_window_size;
    int64_t* offsets = (int64_t*)malloc(sizeof(int64_t) * (size_t)C);
    int rc = ORACLE_OK;
    for (;;) {
        int64_t W = window_size < T ? window_size : T;
        float* table = (float*)malloc(sizeof(float) * (size_t)W * (size_t)C);
        for (int64_t i = 0; i < W * C; ++i) table[i] = (float)cfg->max_prob;
        memset(offsets, 0, sizeof(int64_t) * (size_t)C);
        int64_t t = oracle_fill_table(table, W, C, lpz, T, V, gt, S, offsets, blank, flags);
        int64_t c = C - 1;
        if (cfg->backtrack_from_max_t) t = W - 1;
        if (t_end_out) *t_end_out = t;
        for (int64_t i = 0; i < C; ++i) timings[i] = 0.0;
        for (int64_t i = 0; i < C; ++i) frame_of_label[i] = 0;
        for (int64_t i = 0; i < T; ++i) char_probs[i] = 0.0;
        for (int64_t i = 0; i < T; ++i) state[i] = -2;
        int err = 0;
        int64_t offset = 0;
        while (t != 0 || c != 0) {
            int64_t min_s = -1;
            double min_delta = INFINITY;
            /* max_lpz_prob starts as the Python float max_prob and becomes an
             * np.float32 once a switch_prob replaces it; the scalar type decides
             * whether the residual below is an fp32 or an fp64 subtraction
             * (NumPy 1.24 scalar arithmetic, requirements.txt:51). */
            double max_lpz_prob = cfg->max_prob;
            int max_lpz_is_f32 = 0;
            int64_t cw = np_idx(c, C, &err); /* ground_truth[c, s], offsets[c], table[., c] */
            if (err) break;
            for (int64_t s = 0; s < S; ++s) {
                int64_t g = gt[cw * S + s];
                if (g == -1) continue;
                int64_t pc = np_idx(c - 1 - s, C, &err);
                if (err) break;
                offset = offsets[cw] - ((c - s > 0) ? offsets[pc] : 0);
                double switch_prob;
                int sp_is_f32 = 0;
                if (c > 0) {
                    int64_t r = np_idx(t + offsets[cw], T, &err);
                    int64_t gi = np_idx(g, V, &err);
                    if (err) break;
                    switch_prob = (double)lpz[r * V + gi];
                    sp_is_f32 = 1;
                } else {
                    switch_prob = cfg->max_prob;
                }
                int64_t r0 = np_idx(t, W, &err);
                int64_t r1 = np_idx(t - 1 + offset, W, &err);
                if (err) break;
                float est32 = table[r0 * C + cw] - table[r1 * C + pc]; /* fp32 - fp32 */
                double delta;
                if (sp_is_f32) {
                    float d32 = (float)switch_prob - est32; /* fp32 residual */
                    delta = (double)fabsf(d32);
                } else {
                    delta = fabs(switch_prob - (double)est32);
                }
                if (delta < min_delta) {
                    min_delta = delta;
                    min_s = s;
                }
                if (switch_prob > max_lpz_prob) { /* Python max(max_lpz_prob, switch_prob) */
                    max_lpz_prob = switch_prob;
                    max_lpz_is_f32 = sp_is_f32;
                }
            }
            if (err) break;
            double stay_prob;
            int stay_is_f32 = 0;
            if (t > 0) {
                int64_t r = np_idx(t + offsets[cw], T, &err);
                if (err) break;
                double lb = (double)lpz[r * V + blank];
                /* Python max(lpz[.., blank], max_lpz_prob): first unless second is greater */
                if (max_lpz_prob > lb) {
                    stay_prob = max_lpz_prob;
                    stay_is_f32 = max_lpz_is_f32;
                } else {
                    stay_prob = lb;
                    stay_is_f32 = 1;
                }
            } else {
                stay_prob = cfg->max_prob;
            }
            int64_t r0 = np_idx(t, W, &err);
            int64_t r1 = np_idx(t - 1, W, &err);
            if (err) break;
            float est_stay32 = table[r0 * C + cw] - table[r1 * C + cw];
            double stay_delta;
            if (stay_is_f32) {
                float d32 = (float)stay_prob - est_stay32;
                stay_delta = (double)fabsf(d32);
            } else {
                stay_delta = fabs(stay_prob - (double)est_stay32);
            }
            if (stay_delta > min_delta) {
                /* reverse switch transition */
                if (c > 0) {
                    int64_t fr = np_idx(offsets[cw] + t, T, &err);
                    if (err) break;
                    for (int64_t s = 0; s <= min_s; ++s) {
                        int64_t ci = np_idx(c - s, C, &err);
                        if (err) break;
                        timings[ci] = (double)(offsets[cw] + t) * cfg->index_duration;
                        frame_of_label[ci] = (int32_t)(offsets[cw] + t);
                    }
                    if (err) break;
                    char_probs[fr] = max_lpz_prob;
                    state[fr] = (int32_t)gt[cw * S + min_s];
                }
                c -= 1 + min_s;
                t -= 1 - offset;
            } else {
                int64_t fr = np_idx(offsets[cw] + t, T, &err);
                if (err) break;
                char_probs[fr] = stay_prob;
                state[fr] = -1;
                t -= 1;
            }
        }
        free(table);
        if (err) {
            window_size *= 2;
            if (window_size < cfg->max_window_size) continue;
            rc = ORACLE_INDEX_ERROR;
        }
        break;
    }
    free(offsets);
    return rc;
}

/* np.add.reduce over a contiguous fp64 vector: NumPy's pairwise summation. */
static double np_pairwise_sum(const double* a, int64_t n) {
    if (n < 8) {
        double res = 0.0;
        for (int64_t i = 0; i < n; ++i) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int64_t i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
    }
}

static double np_mean(const double* a, int64_t n) { return np_pairwise_sum(a, n) / (double)n; }

/* Python slice a[lo:hi] of a length-n vector -> [lo', hi') */
static void py_slice(int64_t lo, int64_t hi, int64_t n, int64_t* o_lo, int64_t* o_hi) {
    if (lo < 0) lo += n;
    if (lo < 0) lo = 0;
    if (lo > n) lo = n;
    if (hi < 0) hi += n;
    if (hi < 0) hi = 0;
    if (hi > n) hi = n;
    if (hi < lo) hi = lo;
    *o_lo = lo;
    *o_hi = hi;
}

/*
 * determine_utterance_segments(config, utt_begin_indices, char_probs, timings, text).
 * utt_begin: [U+1].  Outputs seg_start/seg_end/seg_score: [U] fp64.
 * Python's round() is round-half-to-even == rint() in the default FP mode.
 * Returns 0, or ORACLE_INDEX_ERROR when timings[index+1] is out of range.
 */
int oracle_determine_utterance_segments(const oracle_config* cfg, const int64_t* utt_begin,
                                        int64_t U, const double* char_probs, int64_t T,
                                        const double* timings, int64_t C, double* seg_start,
                                        double* seg_end, double* seg_score) {
    const double dur = cfg->index_duration;
    const int64_t n = cfg->score_min_mean_over_L;
    for (int64_t i = 0; i < U; ++i) {
        int err = 0;
        int64_t b = utt_begin[i], e = utt_begin[i + 1];
        int64_t b0 = np_idx(b, C, &err), bm = np_idx(b - 1, C, &err), bp = np_idx(b + 1, C, &err);
        int64_t e0 = np_idx(e, C, &err), em = np_idx(e - 1, C, &err);
        if (err) return ORACLE_INDEX_ERROR;
        double mid_b = (timings[b0] + timings[bm]) / 2;
        double start = pymax(timings[bp] - 0.5, mid_b);
        double mid_e = (timings[e0] + timings[em]) / 2;
        /* Python min(a, b): returns a unless b < a */
        double a_end = timings[em] + 0.5;
        double end = (mid_e < a_end) ? mid_e : a_end;
        int64_t start_t = (int64_t)rint(start / dur);
        int64_t end_t = (int64_t)rint(end / dur);
        double min_avg;
        if (end_t <= start_t) {
            min_avg = -10000000000.0;
        } else if (end_t - start_t <= n) {
            int64_t lo, hi;
            py_slice(start_t, end_t, T, &lo, &hi);
            min_avg = (hi > lo) ? np_mean(char_probs + lo, hi - lo) : NAN;
        } else {
            min_avg = 0.0;
            for (int64_t t = start_t; t < end_t - n; ++t) {
                int64_t lo, hi;
                py_slice(t, t + n, T, &lo, &hi);
                double m = (hi > lo) ? np_mean(char_probs + lo, hi - lo) : NAN;
                if (m < min_avg) min_avg = m; /* Python min(min_avg, m) */
            }
        }
        seg_start[i] = start;
        seg_end[i] = end;
        seg_score[i] = min_avg;
    }
    return ORACLE_OK;
}

/*
 * Convenience: one segment end to end (what CTCSegmentation.get_segments does,
 * reference call site src/iterative_utterance_alignment.py:216).
 */
int oracle_get_segments(const oracle_config* cfg, const float* lpz, int64_t T, int64_t V,
                        const int64_t* gt, int64_t C, int64_t S, const int64_t* utt_begin,
                        int64_t U, double* timings, double* char_probs, int32_t* state,
                        int32_t* frame_of_label, int64_t* t_end, double* seg_start,
                        double* seg_end, double* seg_score) {
    int rc = oracle_ctc_segmentation(cfg, lpz, T, V, gt, C, S, timings, char_probs, state,
                                     frame_of_label, t_end);
    if (rc != ORACLE_OK) return rc;
    return oracle_determine_utterance_segments(cfg, utt_begin, U, char_probs, T, timings, C,
                                               seg_start, seg_end, seg_score);
}

/* Batched driver used by bench.py's cpu_baseline leg (kind "port"): B independent
 * segments with ragged T/C described by offset arrays, single thread. */
int oracle_get_segments_batch(const oracle_config* cfg, int64_t B, const float* lpz,
                              const int64_t* lpz_off, const int32_t* T, int64_t V,
                              const int64_t* gt, const int64_t* gt_off, const int32_t* C,
                              int64_t S, const int64_t* utt_begin, const int64_t* utt_off,
                              const int32_t* U, double* timings, double* char_probs,
                              int32_t* state, int32_t* frame_of_label, int64_t* t_end,
                              double* seg_start, double* seg_end, double* seg_score,
                              int32_t* status) {
    for (int64_t b = 0; b < B; ++b) {
        const int64_t fo = lpz_off[b] / V; /* frame offset of this segment */
        status[b] = oracle_get_segments(
            cfg, lpz + lpz_off[b], T[b], V, gt + gt_off[b] * S, C[b], S, utt_begin + utt_off[b] + b,
            U[b], timings + gt_off[b], char_probs + fo, state + fo, frame_of_label + gt_off[b],
            t_end + b, seg_start + utt_off[b], seg_end + utt_off[b], seg_score + utt_off[b]);
    }
    return 0;
}
