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


# ---- frame sharding (BASELINE config 5: a stream of frames, templates replicated) ------------------------------
def frame_shard(n_frames: int, world: int, rank: int) -> np.ndarray:
    """Frames of a batch owned by ``rank``: rank, rank + world, ...  Frames are independent
    (``Detector::match`` keeps no state between calls, line2Dup.cpp:1078-1150), so nothing is exchanged
    until the per-frame match lists are gathered."""
    return np.arange(rank, n_frames, world, dtype=np.int64)


def all_gather_frame_lists(recs, counts, n_frames: int, group=None):
    """One gather for a frame-sharded batch.

    ``recs``: torch uint8 ``[per_rank * cap * 24]`` — the rank's frames in shard order, ``cap`` records each;
    ``counts``: torch int32 ``[per_rank * 2]`` ({n_matches, overflow} per frame); every rank passes the same
    ``per_rank = ceil(n_frames / world)`` (ranks with one frame less leave the last slot's count at 0).
    Returns a list of ``n_frames`` numpy record arrays in frame order."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    per_rank = counts.numel() // 2
    cap = recs.numel() // (per_rank * MATCH_DTYPE.itemsize)
    g_counts = torch.empty(world * counts.numel(), dtype=counts.dtype, device=counts.device)
    dist.all_gather_into_tensor(g_counts, counts, group=group)
    g_recs = torch.empty(world * recs.numel(), dtype=recs.dtype, device=recs.device)
    dist.all_gather_into_tensor(g_recs, recs, group=group)
    c = g_counts.cpu().numpy().reshape(world, per_rank, 2)
    r = g_recs.cpu().numpy().view(MATCH_DTYPE).reshape(world, per_rank, cap)
    out = []
    for f in range(n_frames):
        rk, slot = f % world, f // world
        if c[rk, slot, 1] != 0 or c[rk, slot, 0] > cap:
            raise RuntimeError(f"match list of frame {f} overflowed on rank {rk}: {c[rk, slot].tolist()} cap={cap}")
        out.append(r[rk, slot, : c[rk, slot, 0]].copy())
    return out


# ---- build sharding (round 3): row bands of the gradient stage -------------------------------------------------
# The reference builds the pyramid serially (line2Dup.cpp:1084-1120) and runs only the template loop in parallel
# (:1166-1170).  Replicating the build on every rank caps the strong scaling of a step at (build + loop) / build, so
# sbm_match_batch_device_banded shards the build as well: rank r computes row band r of every level's orientation map,
# an in-place all-gather assembles the maps, every rank builds the linear memories from them.  These helpers mirror
# the band geometry of the C side (sbm_capi.hip: band_halo, enqueue_pyramid) for tests, bench.py and estimates.
def band_halo(n_levels: int, level: int) -> int:
    """Rows of level ``level`` a rank computes beyond its own band, on either side, so that the next level's band
    (itself widened by its halo) finds its source rows: 0 at the coarsest level, 2 * (halo(l + 1) + 5) below it
    (7x7 Gaussian 3 + Sobel 1 + vote 1 rows of the next level's image, two rows of this level each; the fused
    cv::pyrDown loads its own 5-row window)."""
    e = 0
    for _ in range(n_levels - 2, level - 1, -1):
        e = 2 * (e + 5)
    return e


def band_plan(rows: int, n_levels: int, n_bands: int, band: int) -> List[Tuple[int, int, int, int]]:
    """Per level (own_lo, own_hi, launch_lo, launch_hi): the rows rank ``band`` owns (and contributes to the gather)
    and the rows its gradient launch covers.  Needs rows_l % n_bands == 0 with an even quotient at every level."""
    out = []
    for l in range(n_levels):
        r = rows >> l
        if r % n_bands or (r // n_bands) & 1:
            raise ValueError(f"level {l}: {r} rows do not split into {n_bands} bands of an even number of rows")
        br, e = r // n_bands, band_halo(n_levels, l)
        out.append((band * br, (band + 1) * br, max(0, band * br - e), min(r, (band + 1) * br + e)))
    return out


def all_gather_bands(own_rows, group=None):
    """Assemble a level's orientation map from the ranks' bands: ``own_rows`` is this rank's torch uint8 tensor
    ``[frames, band_rows, cols]``; returns ``[frames, world * band_rows, cols]`` (the layout the library's in-place
    grouped ncclAllGather leaves in HBM)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    parts = [torch.empty_like(own_rows) for _ in range(world)]
    dist.all_gather(parts, own_rows.contiguous(), group=group)
    return torch.cat(parts, dim=1)
