"""Text-side helpers of the caller loops (host logic, negligible cost).

Behaviour follows the reference's helpers so that windows, proportions and utterance lists
come out identical (pinned by tests/golden/anchor_traces.json, which holds the reference's own
outputs):

  split_transcript          <- src/utils/text_utils.py:81-95   (split_long_transcript)
  utterances_for_row        <- src/utils/alignment_utils.py:35-68   (prepare_text)
  joined_length             <- src/utils/alignment_utils.py:80-81   (count_text_length)
  text_to_audio_proportion  <- src/utils/alignment_utils.py:84-106
  aligned_row_count         <- src/utils/alignment_utils.py:71-77   (get_n_aligned_rows)
  shrink_to_fit             <- src/utils/alignment_utils.py:174-196 (find_a_valid_text_to_audio_proportion)
  normalize_transcript      <- src/utils/text_utils.py:49-78
"""
import re
import string

MEAN_PHONEME_S = 0.08   # "mean phoneme duration is 80 ms"
MARGIN = 3              # an utterance may take up to 3x that


def split_transcript(transcript, max_words_sequence=24):
    """Cut a word sequence into consecutive chunks of at most ``max_words_sequence`` words.
    Splitting is on single spaces (so double spaces yield empty "words", as in the reference);
    chunks that strip to nothing are dropped."""
    words = transcript.split(" ")
    chunks = []
    for i in range(int(len(words) / max_words_sequence) + 1):
        piece = " ".join(words[int(i * max_words_sequence):int((i + 1) * max_words_sequence)]).strip()
        if piece:
            chunks.append(piece)
    return chunks


def utterances_for_row(transcript, max_words_sequence=None, min_words_sequence=None):
    """Row transcription -> list of utterances to align.  With no limits the string is
    returned unchanged (the reference's behaviour for falsy limits)."""
    if max_words_sequence or min_words_sequence:
        if max_words_sequence:
            if len(transcript.split(" ")) > max_words_sequence:
                transcript = split_transcript(transcript, max_words_sequence=max_words_sequence)
            else:
                transcript = [transcript]
        if min_words_sequence:
            raise Exception("Min word sequence not implemented")
    return transcript


def joined_length(utterances):
    """Characters of the utterances joined by single spaces."""
    return len(" ".join(utterances))


def text_to_audio_proportion(audio_length, text_length, sample_rate):
    """> 1: more text than the audio can plausibly hold; < 1: more audio than text."""
    return (text_length * MEAN_PHONEME_S * MARGIN) * sample_rate / audio_length


def aligned_row_count(splits, n_aligned_utterances):
    """How many leading rows are fully covered by ``n_aligned_utterances`` aligned utterances,
    given the number of utterances each processed row was split into."""
    drop = 0
    for i in range(1, len(splits)):
        if sum(splits[:-i]) <= n_aligned_utterances:
            drop = i
            break
    return len(splits[:-drop])  # drop == 0 -> splits[:-0] == [] -> 0, as in the reference


def shrink_to_fit(audio_length, utterances, samples_to_frames_ratio):
    """Drop trailing utterances until the text has fewer characters than the audio has
    frames.  Returns (kept, dropped_last_first); if nothing fits, (original, [])."""
    original = utterances
    max_chars = int(audio_length / samples_to_frames_ratio)
    dropped = []
    for _ in range(1, len(utterances) + 1):
        if joined_length(utterances) < max_chars:
            return utterances, dropped
        dropped.append(utterances[-1])
        utterances = utterances[:-1]
    return original, []


_FONT_OPEN = re.compile(r"<font color=\"#[0-9a-fA-F]{6}\">")
_PUNCT = str.maketrans("", "", string.punctuation)


def _numbers_to_words(text, number_to_words):
    numbers = [int(tok) for tok in text.split() if tok.isdigit()]
    for number in numbers:
        text = text.replace(str(number), number_to_words(number))
    return text


def _default_number_to_words(number):
    try:
        from num2words import num2words
    except ImportError as exc:  # the reference needs num2words too (text_utils.py:4)
        raise NotImplementedError("transcript contains digits and num2words is not installed") from exc
    return num2words(number, lang="es")


def normalize_transcript(transcript, number_to_words=None):
    """Strip caption colour tags and punctuation, lower-case, squeeze spaces, spell numbers."""
    text = _FONT_OPEN.sub("", transcript).replace("</font>", "")
    text = text.replace("\n", " ")
    text = text.translate(_PUNCT).lower()
    for mark in "!¡?¿":
        text = text.replace(mark, "")
    text = text.replace("   ", " ").replace("  ", " ")
    return _numbers_to_words(text, number_to_words or _default_number_to_words)
