"""MI355X-native CTC forced-alignment engine (hot path of
ferugit/iterative-pseudo-forced-alignment-ctc).

Import with ``importlib.import_module("iterative-pseudo-forced-alignment-ctc_amd")`` or
through the root-level alias module ``ipfa_amd``.
"""
from . import _native, synthetic  # noqa: F401
from . import anchor, formats, pipelines, sharding, text_prep, time_reference  # noqa: F401
from . import ctc_segmentation  # noqa: F401
from .alignment import CTCSegmentation, CTCSegmentationTask, wav2vec2_frames  # noqa: F401
from .ctc_segmentation import (CtcSegmentationParameters, prepare_text,  # noqa: F401
                               prepare_token_list)

__all__ = ["CTCSegmentation", "CTCSegmentationTask", "CtcSegmentationParameters", "prepare_text",
           "prepare_token_list", "ctc_segmentation", "synthetic"]
