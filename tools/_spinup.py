"""Shared by the timing tools: run `step` for `ms` of wall time before a measurement.

The card idles at a few hundred MHz and needs ~50-100 ms of load to reach its sustained clocks; a
handful of 0.2 ms launches timed from idle measures that ramp (+20 %), not the kernel."""
import time
import torch


def spin(step, ms=150.0, chunk=16):
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        for _ in range(chunk):
            step()
        torch.cuda.synchronize()
