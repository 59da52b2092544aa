"""Stage scripts behind align_utterances.sh / align_words.sh / search_on_speech.sh.

Same command-line flags, TSV columns and file names as the reference's
``src/iterative_utterance_alignment.py`` (:407-505), ``src/word_level_alignment.py`` (:14-166)
and ``src/search_on_speech.py`` (:15-152), so the bash drivers can call
``python -m`` equivalents of these mains unchanged.  What differs is the execution model:

* rows (word level / search) and audio files (utterance level) are independent, so their
  DP requests are batched into single launches of the HIP engine instead of one
  ``get_segments`` call each (word_level_alignment.py:35, search_on_speech.py:45,
  iterative_utterance_alignment.py:436);
* several processes (one per GPU: RANK / WORLD_SIZE / LOCAL_RANK, as torch.distributed.run sets
  them) shard the work deterministically -- files / rows sorted by estimated cost and packed
  greedily onto the ranks (``sharding.assign_units``) -- instead of racing for empty "claim" files
  (iterative_utterance_alignment.py:440-447); the skip-if-result-exists resume rule is kept, a result
  file appears only when it is complete (temporary file + rename), and the row-level stages gather
  fixed-width result records on rank 0 (``sharding.gather_records``), which writes the one TSV.

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
        if num_frames == 0 or num_frames < -1:   # as torchaudio.load: the callers' except branches rely on it
            raise ValueError("Invalid argument: num_frames must be -1 or greater than 0.")
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
def _file_costs(df, paths):
    """Work estimate per audio file: frames x label columns summed over its rows."""
    costs = []
    for path in paths:
        rows = df[df["Sample_Path"] == path]
        dur = (rows["End"].astype(float) - rows["Start"].astype(float)).clip(lower=0.0)
        chars = rows["Transcription"].astype(str).str.len()
        costs.append(float((dur * 50.0 * chars).sum()) + 1.0)
    return costs


def _write_tsv_atomic(path, rows, columns):
    tmp = path + ".tmp.%d" % os.getpid()
    write_tsv(tmp, rows, columns)
    os.replace(tmp, path)   # a result file exists only when it is complete


def align_utterance_files(asr_model, aligner, df, vad_df, dst, logs_path, params, opener=WavFile,
                          rank=0, world=1, files_per_round=32, speculate=1, batch_lpz=False, frames_fn=None):
    """File loop of iterative_utterance_alignment.main (:436-475), ``files_per_round`` files in
    lockstep.  Returns the list of result TSV paths written by this rank.

    Files are packed onto the ranks by estimated cost (every rank computes the same assignment), so
    no claim file is needed; a file whose (non-empty) result TSV exists is skipped, as in the
    reference (:440-443) -- an empty one is what a killed run of the reference leaves behind
    (align_utterances.sh:105-107 deletes those) and is redone.  One failing file costs that file:
    the round it was in is repeated file by file and the error is reported at the end.
    ``speculate``: texts computed ahead with every DP request (``anchor.run_batched``; same results).
    ``batch_lpz``: the windows of a round share one padded encoder forward (``get_lpz_batch``: see its
    note on numerics; off by default, as the reference encodes every window alone, :201)."""
    from . import sharding
    samples_to_frames_ratio = aligner.estimate_samples_to_frames_ratio()
    paths = list(dict.fromkeys(df["Sample_Path"].tolist()))
    mine = sharding.assign_units(_file_costs(df, paths), world)[rank]
    todo = []
    for audio_path in (paths[i] for i in mine):
        out = result_path(dst, audio_path)
        if os.path.isfile(out) and os.path.getsize(out) > 0:
            print("File " + str(out) + " already exist, skipping the alignment generation.")
            continue
        todo.append((audio_path, out))

    def coroutine_for(audio_path):
        rows = df[df["Sample_Path"] == audio_path].reset_index(drop=True).to_dict(orient="records")
        vad_rows = vad_df[vad_df["Sample_Path"] == audio_path].reset_index(drop=True).to_dict(orient="records")
        log = anchor.make_logger(logs_path, audio_path.split("/")[-1].replace(".wav", "")) if logs_path else None
        return anchor.file_alignment(asr_model, opener(audio_path), audio_path, rows, vad_rows,
                                     samples_to_frames_ratio, params, log)

    def finish(audio_path, out, result):
        table = [dict(zip(UTT_COLUMNS, r)) for r in result]
        time_reference.restore_short_scores(table, params.short_utterance_len)
        _write_tsv_atomic(out, table, UTT_COLUMNS)
        written.append(out)

    written, failed = [], []
    for k in range(0, len(todo), files_per_round):
        group = todo[k:k + files_per_round]
        try:
            results = anchor.run_batched([coroutine_for(p) for p, _ in group], aligner, speculate=speculate,
                                         batch_lpz=batch_lpz, frames_fn=frames_fn)
            for (audio_path, out), result in zip(group, results):
                finish(audio_path, out, result)
        except Exception as exc:   # which file it was is not known in a lockstep round: redo it file by file
            print("Alignment round failed ({0}: {1}); repeating its files one at a time.".format(type(exc).__name__, exc))
            for audio_path, out in group:
                if out in written:
                    continue
                try:
                    finish(audio_path, out, anchor.run_batched([coroutine_for(audio_path)], aligner, speculate=speculate,
                                                               batch_lpz=batch_lpz, frames_fn=frames_fn)[0])
                except Exception as one:
                    print("File {0} could not be aligned ({1}: {2}).".format(audio_path, type(one).__name__, one))
                    failed.append((audio_path, one))
    if failed:
        print("{0} file(s) failed on rank {1}: {2}".format(len(failed), rank, [p for p, _ in failed]))
    return written


def _utterance_vocabularies(df, params):
    """What a fixed-width result record refers to by number: audio files, channels, speakers, databases and
    every utterance text the rows can yield -- computed from the input table, identically on every rank."""
    def ids(values):
        return {str(v): k for k, v in enumerate(dict.fromkeys(values))}
    texts = []
    for t in df["Transcription"].tolist():
        u = text_prep.utterances_for_row(str(t).upper(), max_words_sequence=params.max_words_sequence)
        texts.extend([u] if isinstance(u, str) else list(u))
    return {"Sample_Path": ids(df["Sample_Path"].tolist()), "Channel": ids(df["Channel"].tolist()),
            "Speaker_ID": ids(df["Speaker_ID"].tolist()), "Database": ids(df["Database"].tolist()),
            "Transcription": ids(texts)}


def gather_utterance_results(dist, df, dst, tsv_path, params, rank=0, world=1):
    """The utterance stage's one exchange (the role of src/postprocess/merge_aligned_files.py:17-25 without a
    shared results directory): every rank turns the result TSVs of ITS files -- written now or by an earlier,
    resumed run -- into fixed-width float64 records
        (file, row in the file, start and end of the id, channel, length, start, end, score, text, speaker, database)
    (strings by their number in ``_utterance_vocabularies``), the records are all-gathered
    (``sharding.gather_records``: RCCL over xGMI on GPUs), and rank 0 writes ``<name>_aligned.tsv`` -- the file
    ``formats.merge_aligned_files`` makes of the per-file TSVs, byte for byte.  Returns its path on rank 0."""
    import torch

    from . import sharding
    voc = _utterance_vocabularies(df, params)
    paths = list(dict.fromkeys(df["Sample_Path"].tolist()))
    mine = sharding.assign_units(_file_costs(df, paths), world)[rank]
    recs = []
    for f in mine:
        out = result_path(dst, paths[f])
        if not (os.path.isfile(out) and os.path.getsize(out) > 0):
            continue   # not aligned (yet): merge_aligned_files skips such files too
        table = read_tsv(out)
        stem = paths[f].split("/")[-1].replace(".wav", "")
        for k, r in enumerate(table.to_dict(orient="records")):
            id_start, id_end = str(r["Sample_ID"])[len(stem) + 1:].split("_")
            recs.append([float(f), float(k), float(id_start), float(id_end), float(voc["Channel"][str(r["Channel"])]),
                         float(r["Audio_Length"]), float(r["Start"]), float(r["End"]), float(r["Segment_Score"]),
                         float(voc["Transcription"][str(r["Transcription"])]), float(voc["Speaker_ID"][str(r["Speaker_ID"])]),
                         float(voc["Database"][str(r["Database"])])])
    if dist is None:
        rows = sorted(recs, key=lambda r: (r[0], r[1]))
    else:
        local = torch.tensor(recs, dtype=torch.float64).reshape(-1, 12)
        if dist.get_backend() == "nccl":
            local = local.cuda()
        rows = []
        for part in sharding.gather_records(local, dist):
            rows.extend(part.cpu().tolist())
        rows.sort(key=lambda r: (r[0], r[1]))
    if rank != 0 or not rows:
        return None
    back = {k: list(dict.fromkeys(df[k].tolist())) for k in ("Channel", "Speaker_ID", "Database")}
    texts = list(voc["Transcription"])
    table = []
    for f, _, id_start, id_end, ch, length, start, end, score, text, spk, db in rows:
        path = paths[int(f)]
        stem = path.split("/")[-1].replace(".wav", "")
        table.append(["_".join([stem, str(id_start), str(id_end)]), path, back["Channel"][int(ch)], length, start, end, score,
                      texts[int(text)], back["Speaker_ID"][int(spk)], back["Database"][int(db)]])
    out = os.path.join(dst, tsv_path.split("/")[-1].replace(".tsv", "") + "_aligned.tsv")
    _write_tsv_atomic(out, table, UTT_COLUMNS)
    return out


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


def _word_row(row, clip_start, start, end, score):
    """Result row of word_level_alignment.py:112-128 from the kept segment's (offset) times."""
    audio_path = row["Sample_Path"]
    audio_name = audio_path.split("/")[-1]
    extension = audio_name.split(".")[-1]
    abs_start, abs_end = clip_start + start, clip_start + end
    sample_id = "_".join([audio_name.replace(extension, ""), str(abs_start), str(abs_end)])
    return [sample_id, audio_path, end - start, abs_start, abs_end, score, row["Normalized_Transcription"],
            row["Speaker_ID"], row["Wanted_Text"].lower(), row["Database"]]


def _emissions(aligner, waveforms, batch_lpz, frames_fn=None):
    """Log-posteriors of a chunk of clips: one encoder forward each, the reference's way
    (word_level_alignment.py:89, search_on_speech.py:74), or one padded forward for the chunk."""
    if batch_lpz and len(waveforms) > 1 and hasattr(aligner, "get_lpz_batch"):
        return aligner.get_lpz_batch(waveforms, frames_fn)
    return [aligner.get_lpz(w) for w in waveforms]


def word_hits(asr_model, aligner, records, indices, opener=WavFile, time_info=True, offset_time=0.0, left_offset=0.0,
              right_offset=0.0, log=None, rows_per_launch=2048, number_to_words=None, batch_lpz=False, frames_fn=None,
              return_stop=False):
    """Row loop of word_level_alignment.main (:35-135) over ``records[i] for i in indices`` ->
    fixed-width records ``(i, clip_start, start, end, score)``: what a rank hands to the gather.

    Rows go through in chunks of ``rows_per_launch``: load and normalise the clips, encode them, align the
    chunk in ONE launch, keep the records, drop the audio -- a rank never holds more than a chunk of waveforms
    (10 000 rows of a few seconds each are gigabytes).
    ``return_stop``: also return the row at which an unreadable clip ended the run (None: none did) -- the
    reference stops there for good (:63-66); a sharded run drops every hit from that row on (``word_main``)."""
    log = log or (lambda m: None)
    indices = list(indices)
    out, stopped, stop_row = [], False, None
    for k in range(0, len(indices), rows_per_launch):
        chunk = []
        for i in indices[k:k + rows_per_launch]:
            row = records[i]
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
                stopped, stop_row = True, i
                break   # the reference stops the whole run here (:63-66); the rows before it are aligned
            text = sentence_pieces(text_prep.normalize_transcript(row["Normalized_Transcription"], number_to_words).upper(),
                                   row["Wanted_Text"])
            chunk.append((i, row, clip_start, clip_end, waveform, text))
        lpzs = _emissions(aligner, [c[4] for c in chunk], batch_lpz, frames_fn)
        tasks = [aligner.prepare_segmentation_task(text, lpz, row["Sample_ID"], waveform.shape[0])
                 for (_, row, _, _, waveform, text), lpz in zip(chunk, lpzs)]
        for (i, row, clip_start, clip_end, _, _), lines in zip(chunk, _aligned_lines(aligner, tasks)):
            wanted = row["Wanted_Text"]
            if isinstance(lines, AssertionError):
                log(str(lines))
                log("File {0} sequence from {1} to {2} is shorter than text: {3}".format(row["Sample_Path"], clip_start, clip_end, wanted))
                continue
            for seg in lines:
                if len(seg) != 6:
                    log("Some problem with segment: " + str(seg))
                    continue
                if seg[-1] == wanted:
                    start = float(seg[2]) + offset_time + left_offset
                    end = float(seg[3]) + offset_time + right_offset
                    score = float(seg[4])
                    log("{0} | {1} | {2} | {3}".format(round(clip_start + start, 3), round(clip_start + end, 3), round(score, 3), seg[-1]))
                    out.append((float(i), clip_start, start, end, score))
        del chunk, lpzs, tasks   # the clips and emissions of this chunk are not needed again
        if stopped:
            break
    return (out, stop_row) if return_stop else out


def align_words(asr_model, aligner, df, opener=WavFile, time_info=True, offset_time=0.0, left_offset=0.0,
                right_offset=0.0, log=None, rows_per_launch=2048, number_to_words=None, batch_lpz=False):
    """Row loop of word_level_alignment.main (:35-135) -> result rows (WORD_COLUMNS)."""
    records = df.to_dict(orient="records")
    hits = word_hits(asr_model, aligner, records, range(len(records)), opener, time_info, offset_time, left_offset,
                     right_offset, log, rows_per_launch, number_to_words, batch_lpz)
    return [_word_row(records[int(i)], cs, st, en, sc) for i, cs, st, en, sc in hits]


# --------------------------------------------------------------------- search-on-speech stage
def _sos_row(row, wanted_text, clip_start, start, end, score):
    """Result row of search_on_speech.py:97-113."""
    audio_path = row["Sample_Path"]
    audio_name = audio_path.split("/")[-1]
    extension = audio_name.split(".")[-1]
    abs_start, abs_end = clip_start + start, clip_start + end
    sample_id = "_".join([audio_name.replace(extension, ""), str(abs_start), str(abs_end)])
    return [sample_id, audio_path, end - start, abs_start, abs_end, score, row["Speaker_ID"], wanted_text.lower(),
            row["Database"]]


def normalized_query(wanted_text, number_to_words=None):
    if wanted_text == "":
        raise Exception("Sorry, empty text cannot be searched on speech.")
    return text_prep.normalize_transcript(wanted_text, number_to_words).upper()


def search_hits(asr_model, aligner, records, indices, wanted_text, opener=WavFile, offset_time=0.0, left_offset=0.0,
                right_offset=0.0, log=None, rows_per_launch=2048, batch_lpz=False, frames_fn=None):
    """Row loop of search_on_speech.main (:45-120) over ``records[i] for i in indices`` (``wanted_text``
    already normalised) -> records ``(i, clip_start, start, end, score)``; chunked like ``word_hits``."""
    log = log or (lambda m: None)
    query = "·" + wanted_text.strip() + "·"
    indices = list(indices)
    out, waveform, last_loaded = [], None, None
    for k in range(0, len(indices), rows_per_launch):
        chunk = []
        for i in indices[k:k + rows_per_launch]:
            row = records[i]
            clip_start, clip_end = float(row["Start"]), float(row["End"])
            try:
                src = opener(row["Sample_Path"])
                clip, sr = src.load(int(clip_start * src.sample_rate), int((clip_end - clip_start) * src.sample_rate))
                waveform = asr_model.audio_normalizer(clip, sr)
            except Exception:   # the reference prints and goes on with the previous row's audio (:66-67)
                print("Start frame: {0}. Enf frame: {1}. Row: {2}".format(clip_start, clip_end, row))
                if last_loaded != i - 1:
                    # "the previous row" is the previous row of the TABLE -- in a sharded run not the row this rank
                    # aligned before: the clip of the nearest earlier row that can be read
                    waveform = None
                    for k in range(i - 1, -1, -1):
                        try:
                            prev = records[k]
                            src = opener(prev["Sample_Path"])
                            ps, pe = float(prev["Start"]), float(prev["End"])
                            clip, sr = src.load(int(ps * src.sample_rate), int((pe - ps) * src.sample_rate))
                            waveform = asr_model.audio_normalizer(clip, sr)
                            break
                        except Exception:
                            continue
            last_loaded = i
            if waveform is None:   # (the reference has no audio at all at this point and fails)
                raise RuntimeError("row {0}: no readable audio at or before this row".format(i))
            chunk.append((i, row, clip_start, clip_end, waveform))
        lpzs = _emissions(aligner, [c[4] for c in chunk], batch_lpz, frames_fn)
        tasks = [aligner.prepare_segmentation_task(query, lpz, row["Sample_ID"], wf.shape[0])
                 for (_, row, _, _, wf), lpz in zip(chunk, lpzs)]
        for (i, row, clip_start, clip_end, _), lines in zip(chunk, _aligned_lines(aligner, tasks)):
            if isinstance(lines, AssertionError):
                log(str(lines))
                log("File {0} sequence from {1} to {2} is shorter than text: {3}".format(row["Sample_Path"], clip_start, clip_end, wanted_text))
                continue
            for seg in lines:
                if len(seg) != 6:
                    log("Some problem with segment: " + str(seg))
                    continue
                if seg[-1] == query:
                    start = float(seg[2]) + offset_time + left_offset
                    end = float(seg[3]) + offset_time + right_offset
                    score = float(seg[4])
                    log("{0} | {1} | {2} | {3}".format(round(clip_start + start, 3), round(clip_start + end, 3), round(score, 3),
                                                       seg[-1].replace("·", "")))
                    out.append((float(i), clip_start, start, end, score))
        del chunk, lpzs, tasks
    return out


def search_on_speech(asr_model, aligner, df, wanted_text, opener=WavFile, offset_time=0.0, left_offset=0.0,
                     right_offset=0.0, log=None, rows_per_launch=2048, number_to_words=None, batch_lpz=False):
    """Row loop of search_on_speech.main (:45-120) -> result rows (SOS_COLUMNS)."""
    wanted_text = normalized_query(wanted_text, number_to_words)
    records = df.to_dict(orient="records")
    hits = search_hits(asr_model, aligner, records, range(len(records)), wanted_text, opener, offset_time, left_offset,
                       right_offset, log, rows_per_launch, batch_lpz)
    return [_sos_row(records[int(i)], wanted_text, cs, st, en, sc) for i, cs, st, en, sc in hits]


# --------------------------------------------------------------------------------- CLI mains
def _rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def _local_device():
    """One process per GPU: LOCAL_RANK picks the device of this process (torch.distributed.run sets it)."""
    return int(os.environ.get("LOCAL_RANK", "0"))


def _gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def _load_model_and_aligner(args, **aligner_kwargs):
    from speechbrain.pretrained import EncoderASR   # the acoustic model stays SpeechBrain's

    from . import _native
    from .alignment import CTCSegmentation
    run_opts, engine = None, None
    if _gpu_available():
        import torch
        dev = _local_device()
        torch.cuda.set_device(dev)
        run_opts = {"device": "cuda:%d" % dev}
        engine = _native.Engine(dev)   # the DP engine of THIS rank's GPU, not device 0
    asr_model = EncoderASR.from_hparams(source=args.asr_hub, savedir=args.asr_savedir, run_opts=run_opts)
    # model on a GPU: the log-posteriors stay in HBM and feed the DP kernels where they are (no PCIe round
    # trip per window; the reference computes them once per window, iterative_utterance_alignment.py:201)
    return asr_model, CTCSegmentation(asr_model, kaldi_style_text=False, time_stamps="fixed", engine=engine,
                                      keep_lpz_on_device=engine is not None, **aligner_kwargs)


def _process_group():
    """(dist or None, rank, world): joins the default process group when WORLD_SIZE > 1 -- "nccl"
    (= RCCL over xGMI) when this process has a GPU, "gloo" otherwise (CPU tests)."""
    rank, world = _rank_world()
    if world <= 1:
        return None, 0, 1
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if _gpu_available() and os.environ.get("CTCFA_DIST_BACKEND", "nccl") == "nccl":
            torch.cuda.set_device(_local_device())
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", _local_device()))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    return dist, dist.get_rank(), dist.get_world_size()


def _row_costs(records):
    """Work estimate per row: frames x label columns."""
    costs = []
    for r in records:
        try:
            dur = max(0.0, float(r["End"]) - float(r["Start"]))
        except Exception:
            dur = float(r.get("Audio_Length", 1.0) or 1.0)
        text = str(r.get("Normalized_Transcription", r.get("Transcription", "")))
        costs.append(dur * 50.0 * (len(text) + 2) + 1.0)
    return costs


def _gather_hits(dist, hits, world, stop_row=None, error=None):
    """Every rank's (row, clip_start, start, end, score) records -> all of them, in row order (the
    single exchange of the row-level stages; role of src/postprocess/merge_aligned_files.py:17-25).
    ``stop_row``: the row at which this rank's run ended for good (word level: an unreadable clip) -- hits from the
    smallest such row on are dropped, as the reference's single process never reaches them.  ``error``: an
    exception this rank ran into; it still takes part in the collective (nobody is left waiting) and every rank
    raises afterwards."""
    if dist is None:
        if error is not None:
            raise error
        return sorted(h for h in hits if stop_row is None or h[0] < stop_row)
    import torch

    from . import sharding
    on_gpu = dist.get_backend() == "nccl"
    local = torch.tensor(hits, dtype=torch.float64).reshape(-1, 5)
    note = torch.tensor([[float("inf") if stop_row is None else float(stop_row), 1.0 if error is not None else 0.0]],
                        dtype=torch.float64)
    if on_gpu:
        local, note = local.cuda(), note.cuda()
    notes = torch.cat([n.cpu() for n in sharding.gather_records(note, dist)])
    rows = sharding.merge_in_unit_order(sharding.gather_records(local, dist), None)
    if bool((notes[:, 1] > 0).any()):
        raise error if error is not None else RuntimeError("another rank failed in the row stage (see its log)")
    stop = float(notes[:, 0].min())
    return [tuple(r) for r in rows if r[0] < stop]


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
    # not a flag of the reference: shrunk texts aligned ahead with every request (0 = one text per launch)
    p.add_argument("--speculate", type=int, default=1)
    # not flags of the reference either: audio files advanced in lockstep (one launch per round carries a window of
    # each), and whether a round's windows share one padded encoder forward (see CTCSegmentation.get_lpz_batch)
    p.add_argument("--files_per_round", type=int, default=32)
    p.add_argument("--batch_lpz", action="store_true")
    # ... and the merge step of align_utterances.sh done in-process: the ranks gather their result records and rank 0
    # writes <tsv name>_aligned.tsv into --dst (what src/postprocess/merge_aligned_files.py makes of the per-file TSVs)
    p.add_argument("--gather", action="store_true")
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
    gather = bool(getattr(args, "gather", False))
    dist = None
    if gather:
        dist, rank, world = _process_group()
    else:
        rank, world = _rank_world()
    df = read_tsv(df_path)
    written = align_utterance_files(asr_model, aligner, df, read_tsv(vad_path), args.dst,
                                    args.logs_path, params, opener, rank, world,
                                    files_per_round=max(1, int(getattr(args, "files_per_round", 32))),
                                    speculate=getattr(args, "speculate", 1), batch_lpz=bool(getattr(args, "batch_lpz", False)))
    if gather:
        merged = gather_utterance_results(dist, df, args.dst, df_path, params, rank, world)
        if merged:
            print("Aligned partition written to " + merged)
    return written


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
    # not flags of the reference: rows per DP launch (a chunk's clips are loaded, encoded, aligned and dropped
    # together) and one padded encoder forward per chunk instead of one per row
    p.add_argument("--rows_per_launch", type=int, default=2048)
    p.add_argument("--batch_lpz", action="store_true")
    return p


def word_main(args, asr_model=None, aligner=None, opener=WavFile, number_to_words=None):
    """word_level_alignment.main for one process of WORLD_SIZE: this rank aligns its share of the
    rows, the records are gathered, rank 0 writes ``<name>_words.tsv`` (the other ranks return None)."""
    from . import sharding
    if asr_model is None:
        asr_model, aligner = _load_model_and_aligner(args)
    dist, rank, world = _process_group()
    log = anchor.make_logger(args.logs_path, args.tsv_path.split("/")[-1].replace(".tsv", "") + ("" if world == 1 else ".rank%d" % rank)) \
        if args.logs_path else None
    records = read_tsv(args.tsv_path).to_dict(orient="records")
    mine = sharding.assign_units(_row_costs(records), world)[rank]
    hits, stop_row, error = [], None, None
    try:
        hits, stop_row = word_hits(asr_model, aligner, records, mine, opener, args.time_info, args.offset_time, args.left_offset,
                                   args.right_offset, log, rows_per_launch=max(1, int(getattr(args, "rows_per_launch", 2048))),
                                   number_to_words=number_to_words, batch_lpz=bool(getattr(args, "batch_lpz", False)),
                                   return_stop=True)
    except Exception as exc:   # (this rank still joins the gather: the others must not wait for it for ever)
        error = exc
    hits = _gather_hits(dist, hits, world, stop_row, error)
    if rank != 0:
        return None
    out = args.tsv_path.replace("_filtered.tsv", "_words.tsv")
    _write_tsv_atomic(out, [_word_row(records[int(i)], cs, st, en, sc) for i, cs, st, en, sc in hits], WORD_COLUMNS)
    return out


def search_main(args, asr_model=None, aligner=None, opener=WavFile, number_to_words=None):
    """search_on_speech.main, sharded and gathered like ``word_main``; rank 0 writes ``<name>_sos.tsv``."""
    from . import sharding
    wanted = normalized_query(args.text, number_to_words)
    if asr_model is None:
        asr_model, aligner = _load_model_and_aligner(args)
    dist, rank, world = _process_group()
    log = anchor.make_logger(args.logs_path, args.tsv_path.split("/")[-1].replace(".tsv", "") + ("" if world == 1 else ".rank%d" % rank)) \
        if args.logs_path else None
    records = read_tsv(args.tsv_path).to_dict(orient="records")
    mine = sharding.assign_units(_row_costs(records), world)[rank]
    hits, error = [], None
    try:
        hits = search_hits(asr_model, aligner, records, mine, wanted, opener, args.offset_time, args.left_offset,
                           args.right_offset, log, rows_per_launch=max(1, int(getattr(args, "rows_per_launch", 2048))),
                           batch_lpz=bool(getattr(args, "batch_lpz", False)))
    except Exception as exc:
        error = exc
    hits = _gather_hits(dist, hits, world, None, error)
    if rank != 0:
        return None
    out = os.path.join(args.dst_path, args.tsv_path.split("/")[-1].replace(".tsv", "") + "_sos.tsv")
    _write_tsv_atomic(out, [_sos_row(records[int(i)], wanted, cs, st, en, sc) for i, cs, st, en, sc in hits], SOS_COLUMNS)
    return out
