#!/usr/bin/env python3
"""Tuning aid: six equal tiles as a 7-wave workgroup (forced K) vs the balanced 8-wave layout (auto)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "shape_search.py")).read().split("for name, segs in")[0])
for name, segs, K in (("B=512 C=768", uniform(512, 3000, 26, 28), 2), ("B=512 C=1536", uniform(512, 3000, 59, 25), 4),
                      ("B=512 C=700", uniform(512, 3000, 24, 28), 2)):
    print(name, "auto", run(segs, 0), "forced K", run(segs, K), flush=True)
