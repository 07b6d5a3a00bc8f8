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


def embed(img: np.ndarray, rows: int, cols: int, top: int, left: int) -> np.ndarray:
    """Place ``img`` on a black rows x cols canvas (test.cpp:344-353 pads the same way)."""
    shape = (rows, cols) if img.ndim == 2 else (rows, cols, img.shape[2])
    out = np.zeros(shape, np.uint8)
    out[top : top + img.shape[0], left : left + img.shape[1]] = img
    return out
