"""CPU suite: the C-ABI library loads and exports every symbol include/ctcfa.h declares;
without a GPU the compute entries fail loudly (there is no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "ctcfa.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ctcfa_[a-z_]+)\s*\(", text)))


def test_header_symbols_are_exported(pkg):
    lib = pkg._native.load()
    names = _declared()
    assert len(names) >= 14
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(pkg._native.EXPORTS) == names, "python binding list out of sync with include/ctcfa.h"
    assert lib.ctcfa_version() == 100


def test_status_strings_and_defaults(pkg):
    nat = pkg._native
    lib = nat.load()
    assert lib.ctcfa_status_string(1) == b"Audio is shorter than text!"
    p = nat.default_params()
    assert (p.blank, p.flags, p.min_window_size, p.max_window_size, p.score_min_mean_over_L) == (0, 2, 8000, 100000, 30)
    assert p.index_duration == 0.025


def test_no_gpu_means_loud_failure(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pkg._native.NativeLibraryError, match="no CPU fallback|no HIP device"):
        pkg._native.Engine(0)
    config = pkg.CtcSegmentationParameters()
    import numpy as np
    lpz, gt, ub = pkg.synthetic.make_segment(0, 50, 8, 1, 5)
    with pytest.raises(pkg._native.NativeLibraryError):
        pkg.ctc_segmentation.ctc_segmentation(config, lpz, gt.reshape(-1, 1))


def test_missing_library_is_loud(pkg, monkeypatch, tmp_path):
    nat = pkg._native
    monkeypatch.setattr(nat, "_lib", None)
    monkeypatch.setattr(nat, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(nat.NativeLibraryError, match="not built"):
        nat.load()


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under the product package may import,
    load or link it."""
    pkgdir = os.path.join(ROOT, "iterative-pseudo-forced-alignment-ctc_amd")
    banned = ("import oracle", "from oracle", "oracle_c", "liboracle", "ctc_segmentation_twin", "oracle/")
    for dirpath, _, files in os.walk(pkgdir):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                hits = [b for b in banned if b in text]
                assert not hits, (os.path.join(dirpath, f), hits)
