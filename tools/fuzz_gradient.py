#!/usr/bin/env python3
"""Adversarial check of the gradient stage (quantizedOrientations + hysteresisGradient + pyrDown, line2Dup.cpp:218-450) against
the CPU oracle (GPU box; test infrastructure): images built to sit ON the decisions of the stage -- gradient magnitudes at
and around the weak threshold (strict '>'), channels that tie for the maximum magnitude (the lower channel wins, :370-387),
constant gradient vectors on orientation-bin boundaries, 3x3 votes with four / five equal neighbours, saturated
checkerboards, isolated impulses, constant regions next to texture (the kernel's constant-row shortcut) -- for random
geometries, channel counts, weak thresholds (integers, fractions, 0, beyond the largest magnitude), both gradient
kernels and random rows per work item.  Every byte of every level's orientation map must equal the oracle's.
usage: python tools/fuzz_gradient.py [n_cases] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from shape_based_matching_amd import capi  # noqa: E402

WEAKS = [0.0, 0.4, 1.0, 2.0, 7.5, 9.99, 30.0, 30.5, 31.0, 59.9, 60.0, 150.0, 400.0, 1019.9, 1500.0]


def make_image(rs, kind, rows, cols, ch):
    shape = (rows, cols, ch)
    if kind == "noise":
        img = rs.randint(0, 256, shape)
    elif kind == "low_noise":  # magnitudes around the small thresholds
        base, k = int(rs.randint(0, 250)), int(rs.randint(1, 6))
        img = base + rs.randint(0, k + 1, shape)
    elif kind == "rects":  # piecewise constant: strong straight edges, corners, large constant regions
        img = np.full(shape, int(rs.randint(0, 256)))
        for _ in range(int(rs.randint(3, 30))):
            r0, c0 = int(rs.randint(0, rows)), int(rs.randint(0, cols))
            r1, c1 = r0 + int(rs.randint(1, rows)), c0 + int(rs.randint(1, cols))
            img[r0:r1, c0:c1] = rs.randint(0, 256, ch)
    elif kind == "ramp":  # one gradient vector everywhere: bin boundaries, magnitude == threshold, uniform votes
        ay, ax = rs.randint(-6, 7), rs.randint(-6, 7)
        yy, xx = np.mgrid[0:rows, 0:cols]
        v = (128 + (ay * yy + ax * xx) // int(rs.randint(1, 5)))
        img = np.repeat(v[:, :, None], ch, axis=2) + rs.randint(0, 2, (1, 1, ch))
    elif kind == "checker":  # saturated, period 1..4
        p = int(rs.randint(1, 5))
        yy, xx = np.mgrid[0:rows, 0:cols]
        v = (((yy // p) + (xx // p)) & 1) * 255
        img = np.repeat(v[:, :, None], ch, axis=2)
    elif kind == "impulses":
        img = np.full(shape, int(rs.randint(0, 256)))
        n = int(rs.randint(1, 200))
        img[rs.randint(0, rows, n), rs.randint(0, cols, n)] = rs.randint(0, 256, (n, ch))
    elif kind == "half":  # texture beside a constant region, the boundary at a random row / column
        img = rs.randint(0, 256, shape)
        if rs.randint(0, 2):
            img[int(rs.randint(0, rows)):, :] = rs.randint(0, 256, ch)
        else:
            img[:, int(rs.randint(0, cols)):] = rs.randint(0, 256, ch)
        if rs.randint(0, 2):
            img[: int(rs.randint(0, rows)), :] = rs.randint(0, 256, ch)
    else:
        raise ValueError(kind)
    img = np.clip(img, 0, 255).astype(np.uint8)
    if ch == 3:
        tie = int(rs.randint(0, 5))  # channels that tie for the maximum magnitude
        if tie == 1:
            img[:, :, 1] = img[:, :, 0]
        elif tie == 2:
            img[:, :, 2] = img[:, :, 1]
        elif tie == 3:
            img[:, :, 1] = img[:, :, 0]
            img[:, :, 2] = img[:, :, 0]
        elif tie == 4:
            img[:, :, 2] = 255 - img[:, :, 0]  # same magnitude, opposite direction
    return np.ascontiguousarray(img if ch == 3 else img[:, :, 0])


def run(n_cases, seed, verbose=True):
    rs = np.random.RandomState(seed)
    O.build()
    O.lib()
    kinds = ["noise", "low_noise", "rects", "ramp", "checker", "impulses", "half"]
    t0 = time.time()
    n_px = 0
    for case in range(n_cases):
        T = [(4, 8), (4, 8), (4,), (8, 8), (4, 8, 8)][int(rs.randint(0, 5))]
        unit = 2 ** (len(T) - 1) * T[-1]
        rows = unit * int(rs.randint(1, max(2, 600 // unit)))
        cols = unit * int(rs.randint(1, max(2, 1100 // unit)))
        ch = int(rs.choice([1, 3]))
        weak = float(WEAKS[int(rs.randint(0, len(WEAKS)))])
        kind = kinds[int(rs.randint(0, len(kinds)))]
        qmode = str(rs.choice(["auto", "tile", "stream"]))
        hs = int(rs.choice([0, 2, 4, 6, 10, 18, 32]))
        img = make_image(rs, kind, rows, cols, ch)
        mask = None
        if rs.randint(0, 5) == 0:
            mask = (rs.randint(0, 4, (rows, cols)) > 0).astype(np.uint8) * 255
        ctx = capi.Context(T=T, weak_threshold=weak, device_id=0)
        ctx.set_quantize_mode(qmode, hs)
        ctx.build_pyramid(img, mask)
        pyr = O.Pyramid.build(img, list(T), weak, mask=mask)
        desc = (case, T, rows, cols, ch, weak, kind, qmode, hs, mask is not None)
        for l in range(len(T)):
            got, want = ctx.get_quantized(l), pyr.quantized(l)
            if not np.array_equal(got, want):
                bad = np.argwhere(got != want)
                raise AssertionError((desc, l, len(bad), bad[:5].tolist()))
            n_px += want.size
        pyr.free()
        ctx.close()
        if verbose:
            print("ok", desc, flush=True)
    if verbose:
        print(f"{n_cases} cases, {n_px / 1e6:.1f} Mpixels compared, {time.time() - t0:.0f} s", flush=True)
    return n_px


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
