"""Alias: ``import ipfa_amd`` returns the package ``iterative-pseudo-forced-alignment-ctc_amd``
(whose directory name is not a Python identifier).  Use attribute access on it
(``ipfa_amd.CTCSegmentation``); do not import submodules through the alias."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("iterative-pseudo-forced-alignment-ctc_amd")
sys.modules[__name__] = _pkg
