"""Stage scripts behind align_utterances.sh / align_words.sh / search_on_speech.sh.

Same command-line flags, TSV columns and file names as the reference's
``src/iterative_utterance_alignment.py`` (:407-505), ``src/word_level_alignment.py`` (:14-166)
and ``src/search_on_speech.py`` (:15-152), so the bash drivers can call
``python -m`` equivalents of these mains unchanged.  What differs is the execution model:

* rows (word level / search) and audio files (utterance level) are independent, so their
  DP requests are batched into single launches of the HIP engine instead of one
  ``get_segments`` call each (word_level_alignment.py:35, search_on_speech.py:45,
  iterative_utterance_alignment.py:436);
* several processes shard the work deterministically by rank (files[rank::world]) instead of
  racing for empty "claim" files (iterative_utterance_alignment.py:440-447); the
  skip-if-result-exists resume rule is kept.

Audio is read through an ``opener(path)`` -> object with ``num_frames``, ``sample_rate`` and
``load(frame_offset, num_frames)``; the default opener reads PCM WAV with the stdlib
(torchaudio is optional).
"""
import argparse
import os
import wave

import numpy as np

from . import anchor, text_prep, time_reference

UTT_COLUMNS = anchor.RESULT_COLUMNS
WORD_COLUMNS = ["Sample_ID", "Sample_Path", "Audio_Length", "Start", "End", "Segment_Score", "Transcription",
                "Speaker_ID", "Word", "Database"]
SOS_COLUMNS = ["Sample_ID", "Sample_Path", "Audio_Length", "Start", "End", "Segment_Score", "Speaker_ID", "Word",
               "Database"]


# ------------------------------------------------------------------------------------ audio
class WavFile:
    """PCM WAV reader with the two calls the loops need (torchaudio.info / torchaudio.load)."""

    def __init__(self, path):
        self.path = path
        with wave.open(path, "rb") as w:
            self.sample_rate = w.getframerate()
            self.num_frames = w.getnframes()
            self._channels = w.getnchannels()
            self._width = w.getsampwidth()

    def load(self, frame_offset, num_frames):
        import torch
        with wave.open(self.path, "rb") as w:
            w.setpos(min(max(frame_offset, 0), self.num_frames))
            raw = w.readframes(self.num_frames - frame_offset if num_frames < 0 else num_frames)
        dtype = {1: np.uint8, 2: np.int16, 4: np.int32}[self._width]
        data = np.frombuffer(raw, dtype=dtype).reshape(-1, self._channels).astype(np.float32)
        if self._width == 1:
            data = (data - 128.0) / 128.0
        else:
            data /= float(2 ** (8 * self._width - 1))
        return torch.from_numpy(data), self.sample_rate   # [n, channels] (channels_first=False)


# -------------------------------------------------------------------------------------- TSV
def read_tsv(path):
    import pandas as pd
    return pd.read_csv(path, header=0, sep="\t")


def write_tsv(path, rows, columns):
    import pandas as pd
    pd.DataFrame(rows, columns=columns).to_csv(path, sep="\t", index=None)


def result_path(dst, audio_path):
    return os.path.join(dst, audio_path.split("/")[-1].replace(".wav", ".tsv"))


# -------------------------------------------------------------------- utterance-level stage
def align_utterance_files(asr_model, aligner, df, vad_df, dst, logs_path, params, opener=WavFile,
                          rank=0, world=1, files_per_round=8):
    """File loop of iterative_utterance_alignment.main (:436-475), ``files_per_round`` files in
    lockstep.  Returns the list of result TSV paths written by this rank."""
    samples_to_frames_ratio = aligner.estimate_samples_to_frames_ratio()
    todo = []
    for audio_path in list(dict.fromkeys(df["Sample_Path"].tolist()))[rank::world]:
        out = result_path(dst, audio_path)
        if os.path.isfile(out):
            print("File " + str(out) + " already exist, skipping the alignment generation.")
            continue
        open(out, "a").close()   # same marker the reference leaves while a file is in progress
        todo.append((audio_path, out))
    written = []
    for k in range(0, len(todo), files_per_round):
        group = todo[k:k + files_per_round]
        coroutines = []
        for audio_path, _ in group:
            rows = df[df["Sample_Path"] == audio_path].reset_index(drop=True).to_dict(orient="records")
            vad_rows = vad_df[vad_df["Sample_Path"] == audio_path].reset_index(drop=True).to_dict(orient="records")
            log = anchor.make_logger(logs_path, audio_path.split("/")[-1].replace(".wav", "")) if logs_path else None
            coroutines.append(anchor.file_alignment(asr_model, opener(audio_path), audio_path, rows, vad_rows,
                                                    samples_to_frames_ratio, params, log))
        for (audio_path, out), result in zip(group, anchor.run_batched(coroutines, aligner)):
            table = [dict(zip(UTT_COLUMNS, r)) for r in result]
            time_reference.restore_short_scores(table, params.short_utterance_len)
            write_tsv(out, table, UTT_COLUMNS)
            written.append(out)
    return written


# ------------------------------------------------------------------------- word-level stage
def sentence_pieces(normalized_sentence, wanted_text):
    """word_level_alignment.py:69-84: cut the sentence around the wanted text and interleave
    the pieces with "·" separators (plus a closing one)."""
    pieces = normalized_sentence.split(wanted_text)
    pieces.insert(1, wanted_text)
    pieces = [p.strip() for p in pieces if p != ""]
    out = []
    for i in range(1, 2 * len(pieces)):
        out.append("·" if (i + 1) % 2 else pieces[int(i / 2)])
    out.append("·")
    return out


def _aligned_lines(aligner, tasks):
    """tasks -> per task: list of 6-field lines, or the AssertionError the reference would catch."""
    batch_fn = getattr(aligner, "get_segments_batch", None)
    if batch_fn is not None:
        results = batch_fn(tasks)
    else:
        results = []
        for task in tasks:
            try:
                results.append(aligner.get_segments(task))
            except AssertionError as exc:
                results.append(exc)
    out = []
    for task, res in zip(tasks, results):
        if isinstance(res, Exception):
            if not isinstance(res, AssertionError):
                raise res
            out.append(res)
        else:
            task.set(**res)
            out.append(anchor.parse_task_lines(str(task)))
    return out


def align_words(asr_model, aligner, df, opener=WavFile, time_info=True, offset_time=0.0, left_offset=0.0,
                right_offset=0.0, log=None, rows_per_launch=256, number_to_words=None):
    """Row loop of word_level_alignment.main (:35-135) -> result rows (WORD_COLUMNS)."""
    log = log or (lambda m: None)
    records = df.to_dict(orient="records")
    prepared = []
    for row in records:
        audio_path = row["Sample_Path"]
        if time_info:
            clip_start, clip_end = float(row["Start"]), float(row["End"])
            clip_length = clip_end - clip_start
        else:
            clip_start, clip_end = 0.0, float(row["Audio_Length"])
            clip_length = clip_end
        try:
            src = opener(audio_path)
            clip, sr = src.load(int(clip_start * src.sample_rate), int(clip_length * src.sample_rate))
            waveform = asr_model.audio_normalizer(clip, sr)
        except Exception:
            print("Start frame: {0}. Enf frame: {1}. Row: {2}".format(clip_start, clip_end, row))
            print("Ending execution as non-valid audio file has been provided.")
            break   # the reference stops the whole run here (:63-66)
        text = sentence_pieces(text_prep.normalize_transcript(row["Normalized_Transcription"], number_to_words).upper(),
                               row["Wanted_Text"])
        prepared.append((row, clip_start, clip_end, waveform, text))
    out = []
    for k in range(0, len(prepared), rows_per_launch):
        chunk = prepared[k:k + rows_per_launch]
        tasks = []
        for row, _, _, waveform, text in chunk:
            lpz = aligner.get_lpz(waveform)
            tasks.append(aligner.prepare_segmentation_task(text, lpz, row["Sample_ID"], waveform.shape[0]))
        for (row, clip_start, clip_end, _, _), lines in zip(chunk, _aligned_lines(aligner, tasks)):
            audio_path = row["Sample_Path"]
            audio_name = audio_path.split("/")[-1]
            extension = audio_name.split(".")[-1]
            wanted = row["Wanted_Text"]
            if isinstance(lines, AssertionError):
                log(str(lines))
                log("File {0} sequence from {1} to {2} is shorter than text: {3}".format(audio_path, clip_start, clip_end, wanted))
                continue
            for seg in lines:
                if len(seg) != 6:
                    log("Some problem with segment: " + str(seg))
                    continue
                if seg[-1] == wanted:
                    start = float(seg[2]) + offset_time + left_offset
                    end = float(seg[3]) + offset_time + right_offset
                    score = float(seg[4])
                    abs_start, abs_end = clip_start + start, clip_start + end
                    sample_id = "_".join([audio_name.replace(extension, ""), str(abs_start), str(abs_end)])
                    log("{0} | {1} | {2} | {3}".format(round(abs_start, 3), round(abs_end, 3), round(score, 3), seg[-1]))
                    out.append([sample_id, audio_path, end - start, abs_start, abs_end, score,
                                row["Normalized_Transcription"], row["Speaker_ID"], wanted.lower(), row["Database"]])
    return out


# --------------------------------------------------------------------- search-on-speech stage
def search_on_speech(asr_model, aligner, df, wanted_text, opener=WavFile, offset_time=0.0, left_offset=0.0,
                     right_offset=0.0, log=None, rows_per_launch=256, number_to_words=None):
    """Row loop of search_on_speech.main (:45-120) -> result rows (SOS_COLUMNS)."""
    log = log or (lambda m: None)
    if wanted_text == "":
        raise Exception("Sorry, empty text cannot be searched on speech.")
    wanted_text = text_prep.normalize_transcript(wanted_text, number_to_words).upper()
    query = "·" + wanted_text.strip() + "·"
    records = df.to_dict(orient="records")
    prepared, waveform = [], None
    for row in records:
        clip_start, clip_end = float(row["Start"]), float(row["End"])
        try:
            src = opener(row["Sample_Path"])
            clip, sr = src.load(int(clip_start * src.sample_rate), int((clip_end - clip_start) * src.sample_rate))
            waveform = asr_model.audio_normalizer(clip, sr)
        except Exception:   # the reference prints and goes on with the previous row's audio (:66-67)
            print("Start frame: {0}. Enf frame: {1}. Row: {2}".format(clip_start, clip_end, row))
        prepared.append((row, clip_start, clip_end, waveform))
    out = []
    for k in range(0, len(prepared), rows_per_launch):
        chunk = prepared[k:k + rows_per_launch]
        tasks = []
        for row, _, _, wf in chunk:
            lpz = aligner.get_lpz(wf)
            tasks.append(aligner.prepare_segmentation_task(query, lpz, row["Sample_ID"], wf.shape[0]))
        for (row, clip_start, clip_end, _), lines in zip(chunk, _aligned_lines(aligner, tasks)):
            audio_path = row["Sample_Path"]
            audio_name = audio_path.split("/")[-1]
            extension = audio_name.split(".")[-1]
            if isinstance(lines, AssertionError):
                log(str(lines))
                log("File {0} sequence from {1} to {2} is shorter than text: {3}".format(audio_path, clip_start, clip_end, wanted_text))
                continue
            for seg in lines:
                if len(seg) != 6:
                    log("Some problem with segment: " + str(seg))
                    continue
                if seg[-1] == query:
                    start = float(seg[2]) + offset_time + left_offset
                    end = float(seg[3]) + offset_time + right_offset
                    score = float(seg[4])
                    abs_start, abs_end = clip_start + start, clip_start + end
                    sample_id = "_".join([audio_name.replace(extension, ""), str(abs_start), str(abs_end)])
                    log("{0} | {1} | {2} | {3}".format(round(abs_start, 3), round(abs_end, 3), round(score, 3),
                                                       seg[-1].replace("·", "")))
                    out.append([sample_id, audio_path, end - start, abs_start, abs_end, score, row["Speaker_ID"],
                                wanted_text.lower(), row["Database"]])
    return out


# --------------------------------------------------------------------------------- CLI mains
def _load_model_and_aligner(args, **aligner_kwargs):
    from speechbrain.pretrained import EncoderASR   # the acoustic model stays SpeechBrain's

    from .alignment import CTCSegmentation
    run_opts = {"device": "cuda"} if _gpu_available() else None
    asr_model = EncoderASR.from_hparams(source=args.asr_hub, savedir=args.asr_savedir, run_opts=run_opts)
    return asr_model, CTCSegmentation(asr_model, kaldi_style_text=False, time_stamps="fixed", **aligner_kwargs)


def _gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def _rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def utterance_parser():
    p = argparse.ArgumentParser(description="Iterative pseudo-forced alignment algorithm")
    p.add_argument("--tsv", default="")
    p.add_argument("--vad_segments_tsv", default="")
    p.add_argument("--dst", default="")
    p.add_argument("--logs_path", default="")
    p.add_argument("--asr_hub", default="")
    p.add_argument("--asr_savedir", default="")
    p.add_argument("--threshold", type=float, default=-2.0)
    p.add_argument("--short_utterance_len", type=int, default=30)
    p.add_argument("--max_words_sequence", type=int, default=24)
    p.add_argument("--min_words_sequence", type=int, default=None)
    p.add_argument("--max_window_size", type=float, default=70.0)
    p.add_argument("--window_to_stop", type=float, default=500.0)
    p.add_argument("--min_text_to_audio_prop", type=float, default=0.8)
    p.add_argument("--max_text_to_audio_prop_exec", type=int, default=10)
    return p


def utterance_main(args, asr_model=None, aligner=None, opener=WavFile):
    if asr_model is None:
        asr_model, aligner = _load_model_and_aligner(args, scoring_length=30)
    df_path, vad_path = os.path.normpath(args.tsv), os.path.normpath(args.vad_segments_tsv)
    if not (os.path.isfile(df_path) and os.path.isfile(vad_path)):
        print("{0} or {1} file does not exists, please create it.".format(df_path, vad_path))
        return []
    params = anchor.AnchorParams(
        threshold=args.threshold, short_utterance_len=args.short_utterance_len,
        max_words_sequence=args.max_words_sequence, min_words_sequence=args.min_words_sequence,
        max_window_size=args.max_window_size, window_to_stop=args.window_to_stop,
        min_text_to_audio_prop=args.min_text_to_audio_prop,
        max_text_to_audio_prop_exec=args.max_text_to_audio_prop_exec)
    rank, world = _rank_world()
    return align_utterance_files(asr_model, aligner, read_tsv(df_path), read_tsv(vad_path), args.dst,
                                 args.logs_path, params, opener, rank, world)


def word_parser(search=False):
    p = argparse.ArgumentParser(description="word-level segmentation" if not search else "search words in speech")
    if not search:
        p.add_argument("--use_time_info", dest="time_info", action="store_true")
    p.add_argument("--asr_hub", default="")
    p.add_argument("--asr_savedir", default="")
    p.add_argument("--tsv_path", default="")
    p.add_argument("--dst_path", default="")
    p.add_argument("--offset_time", type=float, default=0.0)
    p.add_argument("--left_offset", type=float, default=0.0)
    p.add_argument("--right_offset", type=float, default=0.0)
    p.add_argument("--collar", type=float, default=0.0)
    p.add_argument("--logs_path", default="")
    if search:
        p.add_argument("--text", default="")
    return p


def word_main(args, asr_model=None, aligner=None, opener=WavFile, number_to_words=None):
    if asr_model is None:
        asr_model, aligner = _load_model_and_aligner(args)
    log = anchor.make_logger(args.logs_path, args.tsv_path.split("/")[-1].replace(".tsv", "")) if args.logs_path else None
    rows = align_words(asr_model, aligner, read_tsv(args.tsv_path), opener, args.time_info, args.offset_time,
                       args.left_offset, args.right_offset, log, number_to_words=number_to_words)
    out = args.tsv_path.replace("_filtered.tsv", "_words.tsv")
    write_tsv(out, rows, WORD_COLUMNS)
    return out


def search_main(args, asr_model=None, aligner=None, opener=WavFile, number_to_words=None):
    if args.text == "":
        raise Exception("Sorry, empty text cannot be searched on speech.")
    if asr_model is None:
        asr_model, aligner = _load_model_and_aligner(args)
    log = anchor.make_logger(args.logs_path, args.tsv_path.split("/")[-1].replace(".tsv", "")) if args.logs_path else None
    rows = search_on_speech(asr_model, aligner, read_tsv(args.tsv_path), args.text, opener, args.offset_time,
                            args.left_offset, args.right_offset, log, number_to_words=number_to_words)
    out = os.path.join(args.dst_path, args.tsv_path.split("/")[-1].replace(".tsv", "") + "_sos.tsv")
    write_tsv(out, rows, SOS_COLUMNS)
    return out
