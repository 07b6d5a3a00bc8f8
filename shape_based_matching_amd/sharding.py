"""Template sharding across GPUs and gathering of match lists.

The reference parallelises ``matchClass`` with one OpenMP ``parallel for`` over
templates whose per-thread match vectors are concatenated by a user-defined
reduction (``line2Dup.cpp:1166-1170``).  The multi-GPU analogue: contiguous,
work-balanced template ranges (one per rank), every rank builds the same
linear-memory pyramid, and the per-rank match lists are concatenated by a
two-step all-gather (counts, then padded records) — RCCL over xGMI when the
process group is ``nccl``, gloo on CPU in the tests.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

from .templates import MATCH_DTYPE, TemplateSet


def coarse_work(ts: TemplateSet, rows: int, cols: int, T: Sequence[int]) -> np.ndarray:
    """Per-template byte-adds of the coarse pass: in-bounds features x template_positions
    (``line2Dup.cpp:818-825, 836-837``) at the coarsest level of a rows x cols frame."""
    L = ts.n_levels
    lc = L - 1
    r, c = rows >> lc, cols >> lc
    t = T[lc]
    W, H = c // t, r // t
    out = np.zeros(ts.n_templates, np.int64)
    for i in range(ts.n_templates):
        lv = ts.levels[i, lc]
        wf = int((int(lv["width"]) - 1) / t) + 1  # C integer division truncates toward zero
        hf = int((int(lv["height"]) - 1) / t) + 1
        npos = (H - hf) * W + (W - wf) + 1
        if npos <= 0:
            continue
        f = ts.feats_of(i, lc)
        inb = int(((f["x"] >= 0) & (f["x"] < c) & (f["y"] >= 0) & (f["y"] < r)).sum())
        out[i] = inb * npos
    return out


def partition(work: np.ndarray, n_shards: int) -> List[Tuple[int, int]]:
    """Contiguous (first, count) ranges with near-equal sums of ``work`` (every template
    costs at least 1 so empty-work templates still spread out)."""
    n = len(work)
    w = np.maximum(np.asarray(work, np.float64), 1.0)
    cum = np.concatenate([[0.0], np.cumsum(w)])
    total = cum[-1]
    bounds = [0]
    for s in range(1, n_shards):
        target = total * s / n_shards
        k = int(np.searchsorted(cum, target, side="left"))
        k = min(max(k, bounds[-1]), n)
        bounds.append(k)
    bounds.append(n)
    return [(bounds[i], bounds[i + 1] - bounds[i]) for i in range(n_shards)]


def all_gather_matches(recs, count, group=None):
    """Concatenate every rank's match records.

    ``recs``: torch uint8 tensor ``[cap * 24]`` (sbm_match_rec bytes) on the
    rank's device, ``count``: torch int32 tensor ``[2]`` ({n_matches, overflow}).
    Two collectives: all-gather of the counts, then all-gather of the padded
    record buffers (RCCL has no all-gatherv).  Returns (numpy records, counts).
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    counts = torch.empty(world * count.numel(), dtype=count.dtype, device=count.device)
    dist.all_gather_into_tensor(counts, count, group=group)
    gathered = torch.empty(world * recs.numel(), dtype=recs.dtype, device=recs.device)
    dist.all_gather_into_tensor(gathered, recs, group=group)
    counts_h = counts.cpu().numpy().reshape(world, -1)
    cap = recs.numel() // MATCH_DTYPE.itemsize
    if (counts_h[:, 1] != 0).any() or (counts_h[:, 0] > cap).any():
        raise RuntimeError(f"match list overflow on some rank: counts={counts_h.tolist()} cap={cap}")
    g = gathered.cpu().numpy().view(MATCH_DTYPE).reshape(world, cap)
    out = np.concatenate([g[r, : counts_h[r, 0]] for r in range(world)]) if world else np.zeros(0, MATCH_DTYPE)
    return out, counts_h[:, 0].copy()
