"""Multi-GPU sharding of the alignment path.

The path shards by independent units -- audio files for the utterance-level stage
(/root/reference/src/iterative_utterance_alignment.py:436; the reference itself shards by
file, with racy claim files :440-447), TSV rows for the word-level / search stages
(word_level_alignment.py:35, search_on_speech.py:45), segments for the synthetic DP bench.
One process per GPU; no collective on the data path.  The single exchange is the gather of
the final result records (the role of src/postprocess/merge_aligned_files.py:17-25), done with
``torch.distributed`` (backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU tests).
"""
import numpy as np


def assign_units(costs, world_size):
    """Deterministic longest-first greedy bin packing.  costs: per-unit work estimate
    (e.g. sum of T*C of a file's windows, or the row's frame count).  Returns a list of
    ``world_size`` index lists; every rank computes the same answer from the same input."""
    order = sorted(range(len(costs)), key=lambda i: (-float(costs[i]), i))
    loads = [0.0] * world_size
    bins = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        bins[r].append(i)
        loads[r] += float(costs[i])
    return [sorted(b) for b in bins]


def gather_records(local, dist, group=None):
    """All-gather ragged fixed-width records.  ``local``: 2-D torch tensor [n_local, width] on
    the backend's device.  Returns the list of per-rank tensors (on every rank).  Two
    collectives: counts, then padded payload -- the payload is tiny (tens of bytes per
    utterance), so the cost is latency, not xGMI bandwidth."""
    import torch
    world = dist.get_world_size(group)
    n_local = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    counts = [int(c.item()) for c in counts]
    width = local.shape[1]
    n_max = max(counts + [1])
    padded = torch.zeros(n_max, width, dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    out = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(out, padded, group=group)
    return [o[:c] for o, c in zip(out, counts)]


def merge_in_unit_order(per_rank_records, per_rank_units):
    """Concatenate gathered records back into global unit order.  Records carry the unit id in
    column 0; ``per_rank_units`` is the assignment from ``assign_units``."""
    rows = []
    for recs in per_rank_records:
        rows.extend(np.asarray(recs.cpu()).tolist())
    rows.sort(key=lambda r: r[0])   # stable: keeps the within-unit order each rank produced
    return rows
