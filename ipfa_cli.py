#!/usr/bin/env python3
"""Command-line entry of the restated stage scripts (same flags as the reference's
src/iterative_utterance_alignment.py:480-507, src/word_level_alignment.py:145-160,
src/search_on_speech.py:131-147), one process per GPU under torch.distributed.run:

    python ipfa_cli.py utterance --tsv ... --vad_segments_tsv ... --dst ... --asr_hub ... --asr_savedir ...
    python ipfa_cli.py word      --tsv_path ..._filtered.tsv --use_time_info --asr_hub ... --asr_savedir ...
    python ipfa_cli.py search    --tsv_path ... --dst_path ... --text "mi amor" --asr_hub ... --asr_savedir ...

The acoustic model is SpeechBrain's EncoderASR (not part of this repository); everything after
``get_lpz`` runs on the MI355X engine.
"""
import sys

import ipfa_amd


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] not in ("utterance", "word", "search"):
        raise SystemExit(__doc__)
    stage, rest = argv[0], argv[1:]
    pl = ipfa_amd.pipelines
    if stage == "utterance":
        return pl.utterance_main(pl.utterance_parser().parse_args(rest))
    if stage == "word":
        return pl.word_main(pl.word_parser().parse_args(rest))
    return pl.search_main(pl.word_parser(search=True).parse_args(rest))


if __name__ == "__main__":
    main()
