#!/usr/bin/env python3
"""Overlap statistics of a rocprofv3 --kernel-trace CSV of the pipelined bench: per kernel the mean duration when batches
overlap, the fraction of wall time during which k kernels run at once, the time with no gradient kernel running.
usage: python tools/timeline_stats.py kernel_trace.csv"""
import csv
import sys

import numpy as np

rows = list(csv.DictReader(open(sys.argv[1])))
name_key = "Kernel_Name" if "Kernel_Name" in rows[0] else [k for k in rows[0] if "ernel" in k and "ame" in k][0]
ev = []
for r in rows:
    n = r[name_key]
    short = next((s for s in ("k_quantize_stream", "k_build_lm", "k_similarity_coarse", "k_similarity_local") if s in n), None)
    if short is None:
        continue
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)))
ev.sort()
# the pipelined phase: the longest streak of gradient launches whose grids are those of throughput sizing (the largest
# level-0 grid is latency sizing's; the bench also runs probes and a one-slot kernel pass)
q = [e for e in ev if e[2] == "k_quantize_stream"]
grids = sorted({e[3] for e in q})
pipe = set(g for g in grids if g not in (max(grids),)) if len(grids) > 2 else set(grids)
lat0 = max(grids)
best, cur_start, cur_n, prev = (0, 0, 0), None, 0, None
for i, e in enumerate(q):
    if e[3] == lat0 or e[3] == sorted(grids)[-3 if len(grids) >= 4 else 0] and False:
        if cur_n > best[0]:
            best = (cur_n, cur_start, prev)
        cur_start, cur_n = None, 0
        continue
    if cur_start is None:
        cur_start = e[0]
    cur_n += 1
    prev = e[1]
if cur_n > best[0]:
    best = (cur_n, cur_start, prev)
t_lo, t_hi = best[1], best[2]
t_lo += (t_hi - t_lo) // 10  # skip the fill
t_hi -= (t_hi - t_lo) // 10
win = [e for e in ev if e[0] >= t_lo and e[1] <= t_hi]
span = (t_hi - t_lo) / 1e3
print(f"window {span:.0f} us, {len(win)} kernels")
by = {}
for s, e, n, g in win:
    by.setdefault((n, g), []).append((e - s) / 1e3)
for (n, g), d in sorted(by.items()):
    print(f"  {n:22s} grid {g:8d}  n={len(d):4d}  mean {np.mean(d):7.1f} us  p10 {np.percentile(d, 10):7.1f}  p90 {np.percentile(d, 90):7.1f}")
# concurrency histogram
pts = []
for s, e, n, g in win:
    pts.append((s, 1, n))
    pts.append((e, -1, n))
pts.sort()
cur, last = 0, t_lo
hist = {}
grad, grad_last, no_grad = 0, t_lo, 0.0
for t, d, n in pts:
    hist[cur] = hist.get(cur, 0) + (t - last)
    if grad == 0:
        no_grad += t - last
    last = t
    cur += d
    if n == "k_quantize_stream":
        grad += d
tot = sum(hist.values())
print("kernels running at once: " + ", ".join(f"{k}: {100 * v / tot:.1f} %" for k, v in sorted(hist.items())))
print(f"no gradient kernel running: {100 * no_grad / tot:.1f} % of the window")
nq = len([1 for s, e, n, g in win if n == "k_quantize_stream"]) / 2
print(f"steps in window ~{nq:.0f}: {span / nq:.1f} us per step")
