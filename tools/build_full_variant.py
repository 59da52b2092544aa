#!/usr/bin/env python3
"""Tuning aid: a FULL build (every vocabulary pitch) of libctcfa_hip.so with extra -D macros, compiled the way
__graft_entry__.build() compiles the product (eight hipcc processes side by side):
    python tools/build_full_variant.py bb1full -DCTCFA_BODY_BLOCKS=1     -> variants/bb1full.so
Load with CTCFA_ALLOW_TUNING_BUILD=1 CTCFA_LIB=$PWD/variants/<name>.so."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

name, flags = sys.argv[1], sys.argv[2:]
ge.HIP_FLAGS = ge.HIP_FLAGS + flags
os.makedirs(os.path.join(ROOT, "variants"), exist_ok=True)
ge._compile_library(os.path.join(ROOT, "variants", name + ".so"))
print("built variants/%s.so with %s" % (name, " ".join(flags)))
