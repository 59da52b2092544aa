"""Iterative anchor alignment of one long audio file (host logic).

Restates the behaviour of ``get_file_iterative_segmentation``
(/root/reference/src/iterative_utterance_alignment.py:14-404) as a *coroutine*: the
per-file logic is strictly sequential (window k+1 starts at the anchor accepted in window k,
reference ``new_segment_start`` :83,:249), but files are independent, so the loop below
never calls the aligner itself.  It ``yield``s requests

    ("lpz", waveform)                               -> [T, V] log-posteriors       (ref :201)
    ("segments", utterances, lpz, name, n_samples)  -> list of 6-field lines, or an
                                                       AssertionError instance      (ref :208-219)

and a driver answers them: ``run_sequential`` one at a time (what the reference does), or
``run_batched``, which advances many files in lockstep and puts all their pending DP requests
into ONE launch of the HIP engine (``CTCSegmentation.get_segments_batch``).

The decision tree that accepts an alignment, drops the last utterance and repeats, or falls
back to the previous attempt (ref :263-379) is kept branch for branch, including the way the
result list is trimmed; tests/golden/anchor_traces.json holds the reference's own outputs for
scripted score sequences and pins this file against them.
"""
import logging
from dataclasses import dataclass

from . import text_prep, time_reference

RESULT_COLUMNS = ["Sample_ID", "Sample_Path", "Channel", "Audio_Length", "Start", "End", "Segment_Score",
                  "Transcription", "Speaker_ID", "Database"]


@dataclass
class AnchorParams:
    """Flags of iterative_utterance_alignment.py:498-505 (same names, same defaults)."""
    threshold: float = -2.0
    short_utterance_len: int = 30
    max_words_sequence: int = 24
    min_words_sequence: object = None
    max_window_size: float = 70.0
    window_to_stop: float = 500.0
    min_text_to_audio_prop: float = 0.8
    max_text_to_audio_prop_exec: int = 10


def parse_task_lines(task_text):
    """``str(task)`` -> list of 6-field lists, as the reference parses it (:218-219)."""
    return [line.split(" ", 5) for line in task_text.strip().split("\n")]


def file_alignment(asr_model, audio, audio_path, rows, vad_rows, samples_to_frames_ratio,
                   params=None, log=None):
    """Coroutine; ``return``s the list of result rows (RESULT_COLUMNS order).

    audio: object with ``num_frames``, ``sample_rate`` and ``load(frame_offset, num_frames)``
    -> (tensor [n, channels], sample_rate); rows / vad_rows: lists of dicts (TSV columns).
    """
    p = params or AnchorParams()
    log = log or (lambda msg: None)
    stem = audio_path.split("/")[-1].replace(".wav", "")
    log("Starting iterative alignment for file: " + str(audio_path))

    anchor = None          # end of the last accepted utterance (absolute seconds)
    pending = []           # utterances taken off a window, most recently dropped last
    splits = []            # utterances per processed row
    results = []
    n_segments = len(rows)
    real_audio_length = audio.num_frames / audio.sample_rate
    log("Audio length: " + str(round(real_audio_length, 2)))
    log("Labels length: " + str(round(float(rows[n_segments - 1]["End"]), 2)))

    rows = time_reference.spread_over_speech(rows, vad_rows, real_audio_length, n_segments)
    n_rows = len(rows)
    failures = 0
    # values the reference keeps across iterations when a guarded step fails (:101-109, :147-159)
    proportion = None
    next_is_gap = None
    following = None
    waveform = None
    audio_length = None
    sr = None

    for row_index in range(n_rows):
        row = rows[row_index]
        if row["Type"] == "Non-Speech":
            anchor = float(row["End"])
            continue
        is_last = (row_index + 1) == n_rows
        clip_start = anchor if anchor is not None else float(row["Start"])
        clip_end = float(row["End"])
        clip_length = clip_end - clip_start
        database = row["Database"]
        utterances = text_prep.utterances_for_row(str(row["Transcription"]).upper(),
                                                  max_words_sequence=p.max_words_sequence)
        splits.append(len(utterances))
        if pending:
            utterances = pending[::-1] + utterances
            pending = []
        text_length = text_prep.joined_length(utterances)
        try:
            proportion = text_prep.text_to_audio_proportion(int(clip_length * audio.sample_rate), text_length,
                                                            audio.sample_rate)
        except Exception:
            log("NaN values...")
        if not is_last:
            following = rows[row_index + 1]
            next_is_gap = following["Type"] == "Non-Speech"

        def speech_segment_ending(prop):
            return prop > 10.0 and next_is_gap and abs(float(following["Start"]) - clip_start) > 5.0

        if clip_length >= p.window_to_stop:
            break  # lost: the window grew past the stop size
        if clip_length >= p.max_window_size or speech_segment_ending(proportion):
            log("Recalculating time references, using last anchor as beginning...")
            rows = time_reference.respread_from_anchor(
                rows, vad_rows, real_audio_length - clip_start,
                text_prep.aligned_row_count(splits, len(results)), n_segments, clip_start, log)
            row = rows[row_index]
            clip_end = float(row["End"])
            clip_length = clip_end - clip_start

        try:
            clip, sr = audio.load(int(clip_start * audio.sample_rate), int(clip_length * audio.sample_rate))
            waveform = asr_model.audio_normalizer(clip, sr)
            audio_length = clip.shape[0]
        except Exception:
            log("Start frame: {0}. End frame: {1}.".format(clip_start, clip_end))
        proportion = text_prep.text_to_audio_proportion(audio_length, text_length, sr)
        log("Text to audio proportion: " + str(proportion))

        if not is_last:
            if not (audio_length > 0):
                pending = utterances[::-1]
                continue
            if proportion < p.min_text_to_audio_prop:
                log("Low quantity of text compared to audio, reading more audio and text...")
                pending = utterances[::-1]
                anchor = clip_start
                continue
            if speech_segment_ending(proportion):
                utterances, pending = text_prep.shrink_to_fit(audio_length, utterances, samples_to_frames_ratio)

        try:
            bad = True
            repeat = True
            previous = []          # lines of the previous attempt (one utterance more)
            previous_abs_end = 0.0
            lpz = yield ("lpz", waveform)
            while bad or repeat:
                log("Segment from {0} to {1}".format(clip_start, clip_end))
                lines = yield ("segments", utterances, lpz, row["Sample_ID"], waveform.shape[0])
                if isinstance(lines, AssertionError):
                    raise lines
                for fields in lines:
                    if len(fields) != 6:
                        log("Some problem with segment: " + str(fields))
                        continue
                    seg_text = fields[-1]
                    seg_start, seg_end, score = float(fields[2]), float(fields[3]), float(fields[4])
                    abs_start = clip_start + seg_start
                    abs_end = clip_start + seg_end
                    if len(seg_text) < p.short_utterance_len:
                        score += 2 * p.threshold          # short utterances never anchor
                    if score < p.threshold:
                        bad = True
                    else:
                        bad = False
                        anchor = abs_end
                    log("{0} | {1} | {2} | {3}".format(round(abs_start, 3), round(abs_end, 3), round(score, 3), seg_text))
                    results.append(["_".join([stem, str(abs_start), str(abs_end)]), audio_path, row["Channel"],
                                    seg_end - seg_start, abs_start, abs_end, score, seg_text, row["Speaker_ID"],
                                    database])
                n = len(lines)

                def drop_current():
                    return results[:len(results) - n]

                def keep_current_drop_previous():
                    return results[:len(results) - (2 * n + 1)] + results[len(results) - n:]

                if is_last:
                    bad = repeat = False          # nothing left to read: keep what we have
                elif bad and repeat and not previous:
                    # first attempt(s) end in a bad utterance: shrink the text, or give up the window
                    pending.append(utterances[-1])
                    results = drop_current()
                    if not utterances[:-1]:
                        log("Alignment of empty string cannot be done. Reading more audio to better align.")
                        bad = repeat = False
                        anchor = clip_start
                    else:
                        utterances = utterances[:-1]
                        log("Misalignment detected. Repeating alignment due low score.")
                elif (not bad or previous) and repeat:
                    if previous:
                        prev_score = float(previous[-2][4])
                        if len(previous[-2][-1]) < p.short_utterance_len:
                            prev_score += 2 * p.threshold
                        if score > -1.0 and not (prev_score == score):
                            anchor = abs_end
                            bad = repeat = False
                            results = keep_current_drop_previous()
                            log("Not repeating because last alignment is very nice.")
                        elif prev_score >= score:
                            repeat = bad = False
                            anchor = previous_abs_end
                            pending = pending[:-1]
                            results = drop_current()
                            log("Not improved results keeping previous alignment. Previous score: {0} | "
                                "Current score: {1}".format(prev_score, score))
                        elif not utterances[:-1]:
                            if not bad:
                                anchor = abs_end
                                results = keep_current_drop_previous()
                            else:
                                pending.append(utterances[-1])
                                results = results[:len(results) - (2 * n + 1)]
                                anchor = clip_start
                            bad = repeat = False
                        elif bad:
                            repeat = bad = False
                            anchor = previous_abs_end
                            pending = pending[:-1]
                            results = drop_current()
                            log("Improved results but score is under the threshold. Keeping previous alignment.")
                        else:
                            previous = lines
                            previous_abs_end = abs_end
                            pending.append(utterances[-1])
                            utterances = utterances[:-1]
                            results = keep_current_drop_previous()
                            log("Results have improved. Continuing iteration...")
                    elif score > -1.0:
                        anchor = abs_end
                        bad = repeat = False
                        log("Not repeating because last alignment is very nice.")
                    elif not utterances[:-1]:
                        anchor = abs_end
                        bad = repeat = False
                    else:
                        pending.append(utterances[-1])
                        utterances = utterances[:-1]
                        previous = lines
                        previous_abs_end = abs_end
                        log("Good results. Starting iteration...")
                    log("Not included transcripts: " + str(pending))
            failures = 0
        except AssertionError as exc:
            log(str(exc))
            log("File {0} sequence from {1} to {2} is shorter than text: {3}".format(audio_path, clip_start, clip_end, utterances))
            pending += utterances[::-1]
            failures += 1
            if failures >= p.max_text_to_audio_prop_exec:
                break
    return results


# ----------------------------------------------------------------------------------------
# drivers
# ----------------------------------------------------------------------------------------
def _answer_segments(aligner, request):
    _, utterances, lpz, name, n_samples = request
    try:
        task = aligner.prepare_segmentation_task(utterances, lpz, name, n_samples)
        task.set(**aligner.get_segments(task))
        return parse_task_lines(str(task))
    except AssertionError as exc:
        return exc


def run_sequential(coroutine, aligner):
    """Answer a file coroutine's requests one at a time (the reference's execution order)."""
    try:
        request = next(coroutine)
        while True:
            if request[0] == "lpz":
                request = coroutine.send(aligner.get_lpz(request[1]))
            else:
                request = coroutine.send(_answer_segments(aligner, request))
    except StopIteration as stop:
        return stop.value


def run_batched(coroutines, aligner, batch_lpz=False, frames_fn=None, speculate=0, stats=None):
    """Advance many file coroutines in lockstep; all DP requests pending in a round go to the
    engine in one launch (``aligner.get_segments_batch``).  With ``batch_lpz`` the emission
    requests of a round also share one padded encoder forward (``aligner.get_lpz_batch``; see
    its note on numerics).  Returns the result lists in order.

    ``speculate`` = n > 0: every DP request also asks, in the same launch, for the n texts the
    state machine may ask for next over the same emissions -- the window's text minus its last
    1..n utterances (the repeats of ``iterative_utterance_alignment.py:203,283,352,374``).  They
    share the request's emissions and, being prefixes of its text, its trellis fill
    (``ctcfa_align_batch_shared``); a later request that was foreseen is answered from these results
    without another launch and read-back.  The answers are the ones a launch would give: the same
    ``prepare_segmentation_task`` / ``get_segments`` on the same arguments.
    ``stats`` (dict, optional) counts rounds, DP launches, requests and requests answered ahead."""
    results = [None] * len(coroutines)
    pending = {}
    for i, co in enumerate(coroutines):
        try:
            pending[i] = next(co)
        except StopIteration as stop:
            results[i] = stop.value
    batch_fn = getattr(aligner, "get_segments_batch", None)
    if stats is None:
        stats = {}
    for k in ("rounds", "launches", "requests", "answered_ahead", "tasks"):
        stats.setdefault(k, 0)
    ahead = {}    # file -> (emissions object, name, n_samples, {tuple(utterances): answer})

    def foreseen(i, r):
        hit = ahead.get(i)
        if hit is None or r[0] != "segments" or hit[0] is not r[2] or hit[1] != r[3] or hit[2] != r[4]:
            return None
        return hit[3].get(tuple(r[1]))

    while pending:
        answers = {}
        dp = [(i, r) for i, r in pending.items() if r[0] == "segments"]
        lz = [(i, r) for i, r in pending.items() if r[0] == "lpz"]
        stats["rounds"] += 1
        stats["requests"] += len(dp)
        if batch_lpz and len(lz) > 1 and hasattr(aligner, "get_lpz_batch"):
            for (i, _), lpz in zip(lz, aligner.get_lpz_batch([r[1] for _, r in lz], frames_fn=frames_fn)):
                answers[i] = lpz
        else:
            for i, r in lz:
                answers[i] = aligner.get_lpz(r[1])
        if dp and batch_fn is not None:
            tasks, owners = [], []     # owners: (file, None) = the request itself, (file, key) = a foreseen text
            for i, r in dp:
                try:
                    tasks.append(aligner.prepare_segmentation_task(r[1], r[2], r[3], r[4]))
                    owners.append((i, None))
                except AssertionError as exc:
                    answers[i] = exc
                    continue
                ahead[i] = (r[2], r[3], r[4], {})
                for d in range(1, speculate + 1):
                    if len(r[1]) - d < 1:
                        break
                    shorter = list(r[1][:len(r[1]) - d])
                    try:
                        tasks.append(aligner.prepare_segmentation_task(shorter, r[2], r[3], r[4]))
                        owners.append((i, tuple(shorter)))
                    except AssertionError:
                        break          # (asked for in earnest later, it raises the same way then)
            stats["launches"] += 1 if tasks else 0
            stats["tasks"] += len(tasks)
            for (i, key), task, res in zip(owners, tasks, batch_fn(tasks)):
                if isinstance(res, Exception):
                    if not isinstance(res, AssertionError):
                        if key is None:
                            raise res
                        continue       # a foreseen text the kernels cannot take: not answered ahead
                    ans = res
                else:
                    task.set(**res)
                    ans = parse_task_lines(str(task))
                if key is None:
                    answers[i] = ans
                else:
                    ahead[i][3][key] = ans
        else:
            for i, r in dp:
                answers[i] = _answer_segments(aligner, r)
        nxt = {}
        for i, ans in answers.items():
            try:
                req = coroutines[i].send(ans)
                while True:
                    hit = foreseen(i, req)
                    if hit is None:
                        break
                    stats["requests"] += 1
                    stats["answered_ahead"] += 1
                    req = coroutines[i].send(hit)
                if req[0] != "segments":
                    ahead.pop(i, None)
                nxt[i] = req
            except StopIteration as stop:
                results[i] = stop.value
                ahead.pop(i, None)
        pending = nxt
    return results


def make_logger(logs_path, name, level=logging.DEBUG):
    """Per-file logger to stdout-less file ``<logs_path>/<name>.log`` (format of
    alignment_utils.py:10-32).  Returns a ``log(msg)`` callable."""
    import os
    logger = logging.getLogger("ipfa." + name)
    logger.setLevel(level)
    logger.propagate = False
    logger.handlers.clear()
    handler = logging.FileHandler(os.path.join(logs_path, name + ".log"), mode="w")
    handler.setFormatter(logging.Formatter("%(asctime)s [%(name)s] %(message)s"))
    logger.addHandler(handler)
    return logger.debug
