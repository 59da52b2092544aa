/*
 * ctcfa.h -- C ABI of the MI355X (gfx950) CTC forced-alignment engine.
 *
 * This is the drop-in boundary for the one hot path of
 * ferugit/iterative-pseudo-forced-alignment-ctc: the per-window CTC-segmentation
 * trellis DP + backtrack + utterance scoring that the reference reaches through
 *     aligner.get_segments(task)      src/iterative_utterance_alignment.py:216
 *                                     src/word_level_alignment.py:100
 *                                     src/search_on_speech.py:85
 * i.e. SpeechBrain's CTCSegmentation.get_segments -> ctc_segmentation.ctc_segmentation
 * (-> Cython cython_fill_table) -> determine_utterance_segments
 * (ctc-segmentation==1.7.1, speechbrain==0.5.11; requirements.txt:13,87).
 * Each entry point below names the reference-side interface it replaces.
 *
 * Plain C: opaque handles, raw pointers, sizes.  No torch / numpy types.
 * The library has NO CPU fallback: every compute entry needs a HIP device and
 * returns CTCFA_ERR_HIP otherwise.
 *
 * Batch geometry ("plan"): B independent segments, ragged.
 *   segment b:  lpz      fp32 [T_b, V]   at  lpz     + lpz_off[b]      (elements)
 *               labels   i32  [C_b]      at  labels  + lab_off[b]      (ground_truth_mat[:,0]; [0] == -1)
 *               utt_begin i32 [U_b + 1]  at  utt_begin + utt_off[b] + b
 *   outputs:    frame_of_label i32 [C_b] at  + lab_off[b]   (timings[c] / index_duration)
 *               char_prob      f32 [T_b] at  + frm_off[b]   (frm_off = prefix sum of T)
 *               state          i32 [T_b] at  + frm_off[b]   (label id | -1 "ε" | -2 untouched)
 *               seg_start/end/score f64 [U_b] at + utt_off[b]
 *               t_end, status  i32 [B]
 */
#ifndef CTCFA_H
#define CTCFA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CTCFA_VERSION 100 /* 0.1.0 */

/* return codes of the API calls */
#define CTCFA_OK 0
#define CTCFA_ERR_INVALID 1     /* bad argument / API misuse                       */
#define CTCFA_ERR_HIP 2         /* HIP runtime error or no device                  */
#define CTCFA_ERR_UNSUPPORTED 3 /* valid request outside what the kernels cover    */
#define CTCFA_ERR_NOMEM 4

/* per-segment status[b] (what the reference would have raised) */
#define CTCFA_ST_OK 0
#define CTCFA_ST_AUDIO_SHORTER_THAN_TEXT 1 /* AssertionError("Audio is shorter than text!"),
                                              caught at iterative_utterance_alignment.py:390 */
#define CTCFA_ST_BACKTRACK_FAILED 2        /* IndexError re-raised by ctc_segmentation()      */
#define CTCFA_ST_WINDOWED_UNSUPPORTED 3    /* windowed regime with T*4 bytes > LDS (T > ~40 000 frames) */
#define CTCFA_ST_TEXT_TOO_LONG 4           /* more label columns than one fill workgroup covers
                                              (ctcfa_max_label_columns: 5 485 for a 32-entry vocabulary,
                                              5 119 for the others up to 256, 961 above) with
                                              T <= min_window_size; the other segments of the batch are aligned */
#define CTCFA_ST_INTERNAL 5                /* a wave of the fill kernel gave up waiting for a progress
                                              counter (bounded spins: a bug must not hang the GPU) */
#define CTCFA_ST_TOO_MANY_LABELS 6         /* a narrowed plan (CTCFA_FLAG_TEXTS_OF_31_LABELS, or created with other labels) met
                                              a text that uses more vocabulary entries beside the blank than its ring
                                              takes (31; 62); the other segments are aligned */

/* flags (CtcSegmentationParameters.flags + the backtrack switch) */
#define CTCFA_FLAG_BLANK_TRANSITION_COST_ZERO 1u    /* gratis_blank; forces checkpoint mode: vocab <= 64, or (ctcfa_align_batch*) <= 63 distinct labels per segment */
#define CTCFA_FLAG_PREAMBLE_TRANSITION_COST_ZERO 2u /* default of the package         */
#define CTCFA_FLAG_BACKTRACK_FROM_MAX_T 4u
/* Not a parameter of the package: the caller's promise that no segment's text uses more than 31 vocabulary entries beside
 * the blank (a character model's window does: the reference's 38-token model).  A plan for
 * a vocabulary of 33 .. 256 entries is then NARROWED: trellis fill and backtrack stage the 32 entries each segment looks at
 * (derived from its labels on the device) and run at the pace of a 32-entry vocabulary -- config 3's shape: 0.155 ms a
 * pipelined step with 38 entries instead of 0.158, 0.165 with 200 instead of 0.467 -- and blank_transition_cost_zero is
 * taken above 64 entries.  The results are the un-narrowed plan's.  A segment that breaks the promise gets status
 * CTCFA_ST_TOO_MANY_LABELS.  Ignored for other vocabularies (shared fills narrow like any other plan); entries that get the labels on
 * the host (ctcfa_align_batch*, ctcfa_plan_create_shared with labels) check them and narrow by themselves.
 * CTCFA_NO_NARROW=1 in the environment switches narrowing off. */
#define CTCFA_FLAG_TEXTS_OF_31_LABELS 8u

/* CtcSegmentationParameters fields the DP reads (test_ctc_segmentation.py:20-38 names them). */
typedef struct ctcfa_params {
    int32_t blank;                 /* config.blank, default 0                    */
    uint32_t flags;                /* CTCFA_FLAG_*; the production scripts use 2 */
    int32_t min_window_size;       /* 8000                                       */
    int32_t max_window_size;       /* 100000                                     */
    int32_t score_min_mean_over_L; /* scoring_length=30 (iterative_utterance_alignment.py:418); 1 .. 2^20 frames --
                                      above 128 the utterances are scored by a second, slower kernel       */
    int32_t reserved;
    double index_duration;         /* samples_to_frames_ratio / fs ("fixed" time stamps) */
} ctcfa_params;

typedef struct ctcfa_engine ctcfa_engine; /* one per (process, device); owns a HIP stream */
typedef struct ctcfa_plan ctcfa_plan;     /* batch geometry + device workspace            */

/* launch-shape report (for DESIGN/bench bookkeeping) */
typedef struct ctcfa_plan_info {
    int32_t batch;
    int32_t cols_per_lane;   /* K                                   */
    int32_t waves_per_seg;   /* compute tiles (pipeline stages) per segment */
    int32_t vocab_pitch;     /* LDS row pitch in entries            */
    int32_t lds_bytes;       /* dynamic LDS of the fill kernel      */
    int32_t n_blocks_max;    /* 32-row blocks of the longest segment */
    int64_t workspace_bytes; /* trellis trace words (decision bits / checkpoint rows) + last-column scores */
    int64_t algorithmic_bytes; /* sum_b 4TV + TC/8 + 4T + 8C + 4T */
    int64_t total_frames;
} ctcfa_plan_info;

int ctcfa_version(void);
const char* ctcfa_status_string(int status);

/* How the library was compiled.  0 = the product build.  Anything else is a kernel-tuning build
 * (tools/build_variant.sh) and must never serve results: ablated builds leave out parts of the
 * fill kernel's hand-over (WRONG results, timing only), stamp builds write cycle counts into output
 * buffers.  The Python binding refuses to load such a library unless CTCFA_ALLOW_TUNING_BUILD=1. */
#define CTCFA_BUILD_ABLATED   1   /* -DCTCFA_ABL != 0 */
#define CTCFA_BUILD_STAMPS    2   /* -DCTCFA_STAMP / -DCTCFA_BT_STAMP */
#define CTCFA_BUILD_ONE_PITCH 4   /* -DCTCFA_DEV_VP32_ONLY: vocabularies up to 32 entries only */
#define CTCFA_BUILD_RETUNED   8   /* any other tuning macro off its default */
int ctcfa_build_flags(void);

/* Engine.  device >= 0 selects the HIP device.  One call at a time per engine: it owns a HIP
 * stream and a grow-only device scratch that ctcfa_align_batch reuses from call to call (the
 * reference runs one alignment process per worker: align_utterances.sh:127-137). */
int ctcfa_engine_create(ctcfa_engine** out, int device);
void ctcfa_engine_destroy(ctcfa_engine* eng);
const char* ctcfa_last_error(const ctcfa_engine* eng);
/* Label columns (ground_truth_mat rows) the widest launch shape of the fill kernel covers for this
 * vocabulary: a longer text is status CTCFA_ST_TEXT_TOO_LONG for its segment. */
int ctcfa_max_label_columns(const ctcfa_engine* engine, int32_t vocab);
void ctcfa_default_params(ctcfa_params* p); /* CtcSegmentationParameters defaults, flags = 2 */

/*
 * Plan = the shapes of one batch.  Host arrays: T[B], C[B], U[B] (U may be NULL:
 * no utterance scoring).  Replaces the per-call shape handling of
 * ctc_segmentation(): the `len(ground_truth) > lpz.shape[0]` assertion and the
 * `min(window_size, lpz.shape[0])` window decision become per-segment status.
 * force_cols_per_lane: 0 = launch-shape model, else K in {1,2,3,4,5,6,8,10,12,16}.
 * vocab <= 256: LDS-staged fill kernel (above 80 entries the ring holds the emissions alone and the tiles work out
 * max(blank, label) themselves).  vocab > 256 (sub-word models): gather kernel; needs
 * CTCFA_FLAG_PREAMBLE_TRANSITION_COST_ZERO and C <= 961, else CTCFA_ERR_UNSUPPORTED.
 */
int ctcfa_plan_create(ctcfa_engine* eng, ctcfa_plan** out, const ctcfa_params* params,
                      int32_t batch, int32_t vocab, const int32_t* T, const int32_t* C,
                      const int32_t* U, int32_t force_cols_per_lane);
void ctcfa_plan_destroy(ctcfa_plan* plan);
int ctcfa_plan_get_info(const ctcfa_plan* plan, ctcfa_plan_info* info);

/*
 * Run a plan on DEVICE-resident buffers (layout in the header comment).
 * Replaces, batched:  cython_fill_table + the backtrack loop of ctc_segmentation()
 * + determine_utterance_segments().  `stream` is a hipStream_t (NULL = the
 * default/null stream, e.g. torch's default stream); the call only enqueues work.
 * d_state, d_utt_begin, d_seg_* may be NULL (skipped).
 */
int ctcfa_plan_run_device(ctcfa_plan* plan, const float* d_lpz, const int32_t* d_labels,
                          const int32_t* d_utt_begin, int32_t* d_frame_of_label,
                          float* d_char_prob, int32_t* d_state, double* d_seg_start,
                          double* d_seg_end, double* d_seg_score, int32_t* d_t_end,
                          int32_t* d_status, void* stream);

/*
 * Pipelined variant for back-to-back batches: the fill kernel of this call is enqueued on
 * `stream`, the backtrack kernel on a stream the plan owns, so that the (latency-bound,
 * one-workgroup-per-segment) backtrack of call k overlaps the fill of call k+1.  The plan
 * rotates through four workspaces; a call whose workspace is still being read (the caller is four
 * runs ahead of the GPU) waits for that backtrack ON THE HOST -- nothing but fill kernels ever
 * enters `stream`.  Outputs of a call are complete on `stream` only after
 * ctcfa_plan_flush(plan, stream), which makes `stream` wait for every outstanding backtrack;
 * give consecutive calls different output buffers.  Same arguments as ctcfa_plan_run_device.
 * ctcfa_plan_flush may be given ANY stream (the fill stream, a copy stream, a collective's stream): it
 * orders that stream behind the outstanding backtracks and nothing else -- the plan keeps its own
 * record of which workspace is still being read, so a flush on another stream never lets a later fill
 * overwrite a workspace early.  It does NOT make `stream` wait for fills still queued elsewhere.
  * Where not a single backtrack workgroup finds room on a CU beside the plan's fill workgroups (LDS, registers:
 * vocabularies of 193+ entries, ...), the entry keeps its contract but runs the two kernels one after the other in
 * the caller's stream: side by side they would only slow each other down.
 */
int ctcfa_plan_run_pipelined(ctcfa_plan* plan, const float* d_lpz, const int32_t* d_labels,
                             const int32_t* d_utt_begin, int32_t* d_frame_of_label,
                             float* d_char_prob, int32_t* d_state, double* d_seg_start,
                             double* d_seg_end, double* d_seg_score, int32_t* d_t_end,
                             int32_t* d_status, void* stream);
int ctcfa_plan_flush(ctcfa_plan* plan, void* stream);

/* Kernel timing with HIP events recorded on the stream the kernels run on.
 * set_timing(slots): keep the events of the last `slots` runs (0 = off, the default).  Called again with the same
 * `slots` it only starts over at the first slot (the events stay: a warm-up can pay their first use).
 * set_timing_stride(k): record only every k-th run (an event record is a queue packet between
 * two kernels; k > 1 samples the run durations at a fraction of that cost).  Default 1.
 * get_timings(n, ...): durations [ms] of the last n RECORDED runs, oldest first; synchronises
 * on those runs' end events.  fill_ms / backtrack_ms: float[n], either may be NULL.
 * get_step_intervals(n, ms): float[n - 1], the time from the start of the fill of one recorded run to the start of
 * the fill of the next recorded run, over the last n recorded runs (stride k: k steps of a back-to-back schedule) --
 * per-step samples of a pipelined loop that a short wall-clock region cannot give (its last backtrack is not overlapped). */
int ctcfa_plan_set_timing(ctcfa_plan* plan, int slots);
int ctcfa_plan_set_timing_stride(ctcfa_plan* plan, int stride);
int ctcfa_plan_get_timings(ctcfa_plan* plan, int n, float* fill_ms, float* backtrack_ms);
int ctcfa_plan_get_step_intervals(ctcfa_plan* plan, int n, float* interval_ms);

/*
 * Host-buffer convenience entry: same computation with HOST pointers; uploads,
 * runs, downloads and synchronises.  This is the call the reference-side binding
 * uses in place of `ctc_segmentation(config, lpz, ground_truth_mat)` +
 * `determine_utterance_segments(...)`.
 * Vocabularies above 256 entries (sub-word models): where no segment group looks at more than 256
 * columns (its blank and distinct labels) and the text is long or the flags are not the package's
 * defaults, the call runs on a compact matrix of exactly those columns (labels renumbered inside,
 * `state` reported in the caller's ids); otherwise on the wide-vocabulary fill kernel, which is
 * built for the default flags and up to 961 label columns.  Same for the _resident / _shared entries.
 */
int ctcfa_align_batch(ctcfa_engine* eng, const ctcfa_params* params, int32_t batch, int32_t vocab,
                      const int32_t* T, const int32_t* C, const int32_t* U, const float* lpz,
                      const int32_t* labels, const int32_t* utt_begin, int32_t* frame_of_label,
                      float* char_prob, int32_t* state, double* seg_start, double* seg_end,
                      double* seg_score, int32_t* t_end, int32_t* status);

/*
 * Same call with the EMISSIONS ALREADY ON THE DEVICE (d_lpz: fp32, segments back to back) and
 * everything else in host memory: what `get_segments` costs when `get_lpz` left the encoder output in
 * HBM (alignment.py, keep_lpz_on_device) -- no emission upload, one packed upload of the small inputs,
 * one result download.  `stream` is the hipStream_t the emissions were produced on (NULL = the null
 * stream); the call enqueues there and returns when the results are in the host buffers.
 */
int ctcfa_align_batch_resident(ctcfa_engine* eng, const ctcfa_params* params, int32_t batch, int32_t vocab,
                               const int32_t* T, const int32_t* C, const int32_t* U, const float* d_lpz,
                               const int32_t* labels, const int32_t* utt_begin, int32_t* frame_of_label,
                               float* char_prob, int32_t* state, double* seg_start, double* seg_end,
                               double* seg_score, int32_t* t_end, int32_t* status, void* stream);

/*
 * Segments that SHARE EMISSIONS, and one trellis fill where their texts are prefixes of one another.
 * Replaces the repeat loop of the anchor iteration: after a low-scoring last utterance the reference
 * calls get_segments again with the same `lpz` and the text minus its last utterance
 * (src/iterative_utterance_alignment.py:201 once per window, :203 loop, :283,352,374 drop) -- a
 * new table fill per attempt.  Here the attempts are segments of ONE call:
 *   emission_of[b] = e <= b : segment b uses the emissions of segment e (emission_of[e] == e,
 *                             T[e] == T[b]); `lpz` holds only the blocks of the segments with
 *                             emission_of[b] == b, back to back, in segment order.
 * Members of an emission group whose label sequence is a proper prefix of the group's longest aligned
 * member are served by that member's fill (column c of the trellis depends on columns <= c only): one
 * fill, one backtrack + scoring per member, results identical to separate calls.  Other members
 * (different text, equal length, status != 0, vocab > 256, T > min_window_size, more than 15 prefixes,
 * two prefixes ending in the same lane) are filled by themselves over the shared emissions -- and so
 * is every member when the batch needs fill tiles wider than two columns per lane (texts of more than
 * ~1 700 label columns: only the one- and two-column tiles carry the "watch column" variant of the row
 * loop), silently: the results do not change, only the number of fills (ctcfa_plan_get_sharing).
 * Inputs other than `lpz` and all outputs are laid out exactly as for ctcfa_align_batch (every
 * segment has its own labels, utt_begin, frame_of_label, char_prob, ... regions).
 * lpz_on_device != 0: `lpz` is a device pointer and `stream` the hipStream_t it was produced on
 * (as ctcfa_align_batch_resident); else a host pointer (`stream` ignored).
 */
int ctcfa_align_batch_shared(ctcfa_engine* eng, const ctcfa_params* params, int32_t batch, int32_t vocab,
                             const int32_t* T, const int32_t* C, const int32_t* U, const int32_t* emission_of,
                             const float* lpz, int32_t lpz_on_device, const int32_t* labels,
                             const int32_t* utt_begin, int32_t* frame_of_label, float* char_prob, int32_t* state,
                             double* seg_start, double* seg_end, double* seg_score, int32_t* t_end,
                             int32_t* status, void* stream);

/*
 * LABEL MATRICES (multi-character tokens): `label_matrix` is ground_truth_mat itself, int32
 * [C_b, label_width] row-major per segment, -1 padded -- entry [c, s] = id of the token made of the
 * s + 1 characters ending in column c (ctc_segmentation.prepare_text, the "classic" text converter of
 * SpeechBrain's CTCSegmentation).  A token of s + 1 characters enters column c from column c - (s + 1)
 * in one frame step (cython_fill_table's loop over s; the backtrack's min_s).  label_width in [1,16];
 * every segment goes through the literal, sequential windowed kernel (any T up to ~40 000 frames):
 * exact, not fast.  Otherwise as ctcfa_align_batch (frame_of_label has C_b entries per segment).
 */
int ctcfa_align_batch_spans(ctcfa_engine* eng, const ctcfa_params* params, int32_t batch, int32_t vocab,
                            int32_t label_width, const int32_t* T, const int32_t* C, const int32_t* U,
                            const float* lpz, const int32_t* label_matrix, const int32_t* utt_begin,
                            int32_t* frame_of_label, float* char_prob, int32_t* state, double* seg_start,
                            double* seg_end, double* seg_score, int32_t* t_end, int32_t* status);

/* Plan for the same geometry (ctcfa_plan_run_device / _pipelined then take `d_lpz` with the shared
 * blocks once).  labels: HOST array of all segments' labels back to back, used to check the prefix
 * property; NULL = the caller vouches that members of a group with C[b] < C[longest] are prefixes.
 * With labels given (shared fills or none: emission_of[b] == b throughout is fine), a vocabulary of 33 .. 256 entries
 * whose segments use at most 31 labels each beside the blank gets a NARROWED plan (CTCFA_FLAG_TEXTS_OF_31_LABELS, which
 * asks for one without the labels); a vocabulary of 65 .. 256 entries whose texts use up to 62 a narrowed plan with a ring
 * of 64 entries (the 64-entry kernels: checkpoint mode, blank_transition_cost_zero).  Such a plan serves any later labels
 * that keep to 31 (62) entries per text; a text that does not gets status CTCFA_ST_TOO_MANY_LABELS. */
int ctcfa_plan_create_shared(ctcfa_engine* eng, ctcfa_plan** out, const ctcfa_params* params, int32_t batch,
                             int32_t vocab, const int32_t* T, const int32_t* C, const int32_t* U,
                             const int32_t* emission_of, const int32_t* labels, int32_t force_cols_per_lane);
/* how many trellis fills / emission blocks the plan's batch needs (either pointer may be NULL) */
int ctcfa_plan_get_sharing(const ctcfa_plan* plan, int32_t* n_fills, int32_t* n_emission_blocks);

#ifdef __cplusplus
}
#endif
#endif /* CTCFA_H */
