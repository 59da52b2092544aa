#!/usr/bin/env python3
"""Tuning aid: tools/shape_search.py for another vocabulary size (argv[1], default 76)."""
import os, sys
V = int(sys.argv[1]) if len(sys.argv) > 1 else 76
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "shape_search.py")).read()
src = src.replace("make_segment(s, T, 32, U, n)", f"make_segment(s, T, {V}, U, n)").replace(".to_native(), 32, T, C, U", f".to_native(), {V}, T, C, U")
src = src.split("for name, segs in")[0] + '''
for name, segs in (("B=512 C=640 V=%d" % V, uniform(512, 3000, 22, 28)),):
    out = []
    for K in (0, 1, 2, 3, 4, 5, 6, 8, 10):
        r = run(segs, K)
        if r:
            out.append(f"{'auto' if K == 0 else 'K'}{r[0]}/W{r[1]}:{r[2]:.0f}")
    print(name, " ".join(out), flush=True)
'''
exec(src.replace("V = int", "V_ = int", 0))
