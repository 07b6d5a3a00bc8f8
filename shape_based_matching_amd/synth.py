"""Seeded synthetic inputs for parity tests and benchmarks.

Stage-B inputs bypass the gradient stage: per pyramid level a sparse one-hot
orientation map plus random templates, some of them planted so that true
positives exist and candidate counts stay small.  Stage-A inputs are images
for the whole match() path.  Everything is driven by ``numpy.random.RandomState``
(MT19937, bit-stable across numpy versions).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

from .templates import TemplateSet, from_pyramids


def onehot_map(rs: np.random.RandomState, rows: int, cols: int, density_permille: int = 20) -> np.ndarray:
    """rows x cols u8, border ring 0, interior non-zero with prob density/1000, one random bit."""
    hit = rs.randint(0, 1000, size=(rows, cols)) < density_permille
    bit = rs.randint(0, 8, size=(rows, cols))
    q = np.where(hit, (1 << bit), 0).astype(np.uint8)
    q[0, :] = 0
    q[-1, :] = 0
    q[:, 0] = 0
    q[:, -1] = 0
    return q


def stage_b(seed: int, rows: int, cols: int, T: Sequence[int], n_templates: int, nf: Sequence[int],
            templ_size: int = 260, plant_every: int = 40, density_permille: int = 20,
            class_id: str = "synth") -> Tuple[List[np.ndarray], TemplateSet]:
    """One-hot maps per level + template set.  nf[l] features at level l;
    template box ``templ_size >> l``; every ``plant_every``-th template is
    written into the maps at a random location (even coordinates)."""
    rs = np.random.RandomState(seed)
    L = len(T)
    maps = [onehot_map(rs, rows >> l, cols >> l, density_permille) for l in range(L)]
    pyramids = []
    for t in range(n_templates):
        tp = []
        for l in range(L):
            w = templ_size >> l
            f = np.stack([rs.randint(0, w + 1, nf[l]), rs.randint(0, w + 1, nf[l]), rs.randint(0, 8, nf[l])], axis=1)
            tp.append({"width": w, "height": w, "tl_x": 0, "tl_y": 0, "pyramid_level": l, "features": f})
        if plant_every and t % plant_every == 0:
            margin = 64
            span_x = cols - templ_size - 2 * margin
            span_y = rows - templ_size - 2 * margin
            if span_x > 0 and span_y > 0:
                step = 1 << L  # keeps the location integral at every level
                px = margin + (rs.randint(0, span_x) // step) * step
                py = margin + (rs.randint(0, span_y) // step) * step
                for l in range(L):
                    f = tp[l]["features"]
                    maps[l][(py >> l) + f[:, 1], (px >> l) + f[:, 0]] = (1 << f[:, 2]).astype(np.uint8)
        pyramids.append(tp)
    for m in maps:  # keep the border ring empty, as hysteresisGradient does
        m[0, :] = 0
        m[-1, :] = 0
        m[:, 0] = 0
        m[:, -1] = 0
    return maps, from_pyramids(pyramids, class_id)


def _template_features(seed: int, t: int, L: int, nf: Sequence[int], templ_size: int) -> List[np.ndarray]:
    """Features of template ``t`` of a ``stage_b_fixed`` set: its own MT19937 stream (seeded by (seed, t)), so any
    rank can generate any contiguous shard without drawing the templates in front of it."""
    rs = np.random.RandomState([seed & 0x7FFFFFFF, t])
    out = []
    for l in range(L):
        w = templ_size >> l
        f = np.empty((nf[l], 3), np.int32)
        f[:, 0] = rs.randint(0, w + 1, nf[l])
        f[:, 1] = rs.randint(0, w + 1, nf[l])
        f[:, 2] = rs.randint(0, 8, nf[l])
        out.append(f)
    return out


def stage_b_fixed(seed: int, rows: int, cols: int, T: Sequence[int], n_templates: int, nf: Sequence[int],
                  templ_size: int = 260, n_plants: int = 32, density_permille: int = 20, first: int = 0,
                  count: int = -1, class_id: str = "synth") -> Tuple[List[np.ndarray], TemplateSet]:
    """Stage-B inputs whose map density does not grow with the template count (SURVEY 8d generator, fixed number of
    plants): the same sparse one-hot maps, ``n_templates`` random templates, and exactly ``min(n_plants, n_templates)``
    of them -- spread evenly over the list, so every contiguous shard holds some -- written into the maps at one
    location each.  ``stage_b`` plants every k-th template: at BASELINE config 4's 36 000 templates x 8191 features
    that saturates the maps (every position becomes a candidate); here the maps keep their ~2 % density whatever
    ``n_templates`` is.  Returns the maps (identical for every shard) and the templates ``[first, first + count)``
    with ``template_id`` = index in the full list."""
    rs = np.random.RandomState(seed)
    L = len(T)
    maps = [onehot_map(rs, rows >> l, cols >> l, density_permille) for l in range(L)]
    n_plants = min(n_plants, n_templates)
    planted = [(k * n_templates) // n_plants for k in range(n_plants)] if n_plants else []
    margin, step = 64, 1 << L
    span_x, span_y = cols - templ_size - 2 * margin, rows - templ_size - 2 * margin
    for t in planted:
        feats = _template_features(seed, t, L, nf, templ_size)
        if span_x <= 0 or span_y <= 0:
            continue
        px = margin + (rs.randint(0, span_x) // step) * step
        py = margin + (rs.randint(0, span_y) // step) * step
        for l in range(L):
            f = feats[l]
            maps[l][(py >> l) + f[:, 1], (px >> l) + f[:, 0]] = (1 << f[:, 2]).astype(np.uint8)
    for m in maps:
        m[0, :] = 0
        m[-1, :] = 0
        m[:, 0] = 0
        m[:, -1] = 0
    if count < 0:
        count = n_templates - first
    from .templates import FEATURE_DTYPE, LEVEL_DTYPE

    per = int(sum(nf))
    levels = np.zeros((count, L), LEVEL_DTYPE)
    xyl = np.empty((count * per, 3), np.int32)
    off = 0
    for i in range(count):
        fl = _template_features(seed, first + i, L, nf, templ_size)
        for l in range(L):
            xyl[off : off + nf[l]] = fl[l]
            off += nf[l]
    lev_off = np.concatenate([[0], np.cumsum(nf)[:-1]]).astype(np.int64)
    for l in range(L):
        levels["width"][:, l] = levels["height"][:, l] = templ_size >> l
        levels["pyramid_level"][:, l] = l
        levels["n_features"][:, l] = nf[l]
        levels["feature_offset"][:, l] = np.arange(count, dtype=np.int64) * per + lev_off[l]
    feats = np.zeros(count * per, FEATURE_DTYPE)
    feats["x"], feats["y"], feats["label"] = xyl[:, 0], xyl[:, 1], xyl[:, 2]
    del xyl
    ts = TemplateSet(L, levels, feats, np.zeros(count, np.int32), np.arange(first, first + count, dtype=np.int32), [class_id])
    return maps, ts


def scene_gray(seed: int, rows: int, cols: int, n_shapes: int = 40) -> np.ndarray:
    """Black background + filled rectangles/ellipses (64..255) + noise in [-2, 2]."""
    rs = np.random.RandomState(seed)
    img = np.zeros((rows, cols), np.int32)
    yy, xx = np.mgrid[0:rows, 0:cols]
    for _ in range(n_shapes):
        cx, cy = rs.randint(0, cols), rs.randint(0, rows)
        a, b = rs.randint(8, max(9, cols // 6)), rs.randint(8, max(9, rows // 6))
        val = rs.randint(64, 256)
        if rs.randint(0, 2):
            m = (np.abs(xx - cx) <= a) & (np.abs(yy - cy) <= b)
        else:
            m = ((xx - cx) / float(a)) ** 2 + ((yy - cy) / float(b)) ** 2 <= 1.0
        img[m] = val
    img += rs.randint(0, 5, size=img.shape) - 2
    return np.clip(img, 0, 255).astype(np.uint8)


def scene_bgr(seed: int, rows: int, cols: int, n_shapes: int = 40) -> np.ndarray:
    return np.stack([scene_gray(seed + k, rows, cols, n_shapes) for k in range(3)], axis=2)


def scene_with_object(seed: int, rows: int, cols: int, obj: np.ndarray) -> np.ndarray:
    """SURVEY 8d's Stage-A scene (random filled shapes + noise in [-2, 2] on every pixel) with ``obj`` pasted at the
    centre: no constant region anywhere, and -- when ``obj`` shows the trained object -- matches exist."""
    out = scene_bgr(seed, rows, cols) if obj.ndim == 3 else scene_gray(seed, rows, cols)
    top, left = (rows - obj.shape[0]) // 2, (cols - obj.shape[1]) // 2
    out[top : top + obj.shape[0], left : left + obj.shape[1]] = obj
    return out


def embed(img: np.ndarray, rows: int, cols: int, top: int, left: int) -> np.ndarray:
    """Place ``img`` on a black rows x cols canvas (test.cpp:344-353 pads the same way)."""
    shape = (rows, cols) if img.ndim == 2 else (rows, cols, img.shape[2])
    out = np.zeros(shape, np.uint8)
    out[top : top + img.shape[0], left : left + img.shape[1]] = img
    return out
