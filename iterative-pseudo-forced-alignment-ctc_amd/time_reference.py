"""Time references of a file's rows (host logic).

The source TSV's Start/End are ignored: every row gets a share of the VAD speech time
proportional to its text length, and a 'Non-Speech' row is inserted wherever the running
time crosses a VAD gap.  Rows are plain dicts (TSV columns + Text_Length + Type).

  spread_over_speech   <- src/utils/alignment_utils.py:118-171  (fix_time_reference)
  respread_from_anchor <- src/utils/alignment_utils.py:199-274  (fix_text_to_time_proportion)
  restore_short_scores <- src/utils/alignment_utils.py:277-281  (remove_artefacts)

Reference quirks that are kept on purpose, because downstream windows depend on them
(tests/golden/anchor_traces.json pins them against the reference's own output):
  * insertion points of the Non-Speech rows are the ORIGINAL row indices + 1, not shifted by
    earlier insertions;
  * respread_from_anchor builds its Non-Speech rows with the value order of
    alignment_utils.py:256-260, which does not match the column order of insert_row
    (:111): Start receives the gap length, End the gap start, Transcription the gap end;
  * respread_from_anchor compares the running (absolute) time with the REMAINING audio
    length when it pins the last row's End.
"""

import numpy as np

# positional column order used when a Non-Speech row is built from a value list
INSERT_COLUMNS = ["Sample_ID", "Sample_Path", "Audio_Length", "Start", "End", "Transcription",
                  "Speaker_ID", "Database", "Channel", "Text_Length", "Type"]


def _insert(rows, idx, values):
    """Insert a row at position idx.  A list is mapped positionally onto INSERT_COLUMNS, a
    dict is taken by key (pandas builds a Series from a Series by label)."""
    row = dict(zip(INSERT_COLUMNS, values)) if isinstance(values, (list, tuple)) else dict(values)
    return rows[:idx] + [row] + rows[idx:]


def _spread(rows, indices, n_segments, vad_rows, speech_length, total_text, start_time, last_row_end):
    """Common sweep: the rows at ``indices`` get consecutive spans of the speech time; the
    running time jumps over a VAD gap whenever it reaches the end of the current speech
    segment.  Returns the indices after which a gap was crossed."""
    acc = start_time
    vad_i = 0
    crossed = []
    for index in indices:
        row = rows[index]
        span = row["Text_Length"] / total_text * speech_length
        row["Start"] = acc
        row["End"] = acc + span
        acc += span
        speech_end = float(vad_rows[vad_i]["End"])
        if acc >= speech_end:
            row["End"] = speech_end
            if vad_i + 1 < len(vad_rows):
                acc = float(vad_rows[vad_i + 1]["Start"])
                vad_i += 1
                crossed.append(index)
        if index + 1 == n_segments and acc < last_row_end:
            row["End"] = last_row_end
    return crossed


def spread_over_speech(rows, vad_rows, real_audio_length, n_segments):
    """rows: the file's TSV rows (dicts).  Returns a new list with Text_Length/Type filled,
    Start/End recomputed and Non-Speech rows inserted at VAD gaps."""
    rows = [dict(r) for r in rows]
    for r in rows:
        r["Text_Length"] = len(str(r["Transcription"]))
    total_text = sum(r["Text_Length"] for r in rows)
    # pandas sums with NumPy (pairwise for >= 8 terms): keep the same summation order
    speech_length = float(np.sum(np.asarray([v["Segment_Length"] for v in vad_rows], dtype=np.float64)))
    crossed = _spread(rows, range(len(rows)), n_segments, vad_rows, speech_length, total_text, 0.0,
                      real_audio_length)
    for r in rows:
        r["Type"] = "Speech"
    # the reference reads these from one sampled row; they are constant within a file
    path, channel, database = str(rows[-1]["Sample_Path"]), int(rows[-1]["Channel"]), str(rows[-1]["Database"])
    for i in range(len(crossed)):
        gap_start = float(vad_rows[i]["End"])
        gap_end = float(vad_rows[i + 1]["Start"])
        rows = _insert(rows, crossed[i] + 1,
                       ["Non-speech-" + str(i), path, gap_end - gap_start, gap_start, gap_end, "Non-Speech",
                        "Non-Speech", database, channel, 0, "Non-Speech"])
    return rows


def respread_from_anchor(rows, vad_rows, remaining_audio, n_aligned, n_segments, last_anchor_time, log=None):
    """Re-distribute the not-yet-aligned rows over the speech time after ``last_anchor_time``
    (used when a window grows too large or a speech segment is about to end)."""
    total_text = sum(r["Text_Length"] for r in rows[n_aligned:])
    vad = [dict(v) for v in vad_rows if v["End"] > last_anchor_time]
    vad[0]["Start"] = last_anchor_time
    for v in vad:
        v["Segment_Length"] = v["End"] - v["Start"]
    speech_length = float(np.sum(np.asarray([v["Segment_Length"] for v in vad], dtype=np.float64)))
    if log:
        log('Remaining audio: {0} | Remaining text: {1} | Remaining speech length {2}'.format(
            remaining_audio, total_text, speech_length))
        log('Aligned index: {0}, Total segments: {1}'.format(n_aligned, n_segments))
    old_gaps = [(i, dict(r)) for i, r in enumerate(rows) if r["Type"] == "Non-Speech"]
    rows = [dict(r) for r in rows if r["Type"] == "Speech"]
    crossed = _spread(rows, range(n_aligned, n_segments), n_segments, vad, speech_length, total_text,
                      last_anchor_time, remaining_audio)
    for r in rows:
        r["Type"] = "Speech"
    path, channel, database = str(rows[-1]["Sample_Path"]), int(rows[-1]["Channel"]), str(rows[-1]["Database"])
    for i in range(len(crossed)):
        gap_start = float(vad[i]["End"])
        gap_end = float(vad[i + 1]["Start"])
        # value order of the reference (alignment_utils.py:256-260), mapped onto INSERT_COLUMNS
        rows = _insert(rows, crossed[i] + 1,
                       ["Non-speech-" + str(i), path, channel, gap_end - gap_start, gap_start, gap_end,
                        "Non-Speech", "Non-Speech", database, 0, "Non-Speech"])
    if len(old_gaps) > len(crossed):
        for k in range(len(old_gaps) - len(crossed)):
            index, row = old_gaps[k]
            rows = _insert(rows, index, row)
    return rows


def restore_short_scores(result_rows, short_utterance_len, score_key="Segment_Score", text_key="Transcription"):
    """Short utterances were penalised by 2*threshold (= -4 with the default threshold) so
    that they never become anchors; give the 4.0 back in the final table."""
    for r in result_rows:
        if len(r[text_key]) < short_utterance_len:
            r[score_key] = r[score_key] + 4.0
    return result_rows
