#!/usr/bin/env python3
"""Derivation of the integer orientation-binning rule used by k_quantize (run in the authoring container).

hysteresisGradient bins the phase image with saturate_cast<uchar>(cvRound(fastAtan2(dy, dx) * 16/360))
(line2Dup.cpp:225, :327).  For integer Sobel gradients |dx|, |dy| <= 1020 that float pipeline is a step
function of min/max per octant; this script scans every (max, min) pair per octant path with the CPU
oracle, finds the interval of rational thresholds that reproduces each step, picks the simplest fraction
inside (Stern-Brocot) and verifies the resulting integer rule on all 2041^2 gradient pairs.
Result: cls = (min*367 >= 73*max) + (min*395 >= 264*max); k = cls; if |dy| > |dx|: k = 4-k;
if dx < 0: k = 8-k; if dy < 0: k = 16-k.  0 mismatches.
"""
import sys, numpy as np
from fractions import Fraction
sys.path.insert(0,'/root/repo')
from oracle import oracle as O
M = 1020
res = {}
for s in (0,1):
  for xn in (0,1):
    for yn in (0,1):
        lower = [Fraction(0), Fraction(0)]; upper=[Fraction(10), Fraction(10)]
        vals=set(); ok=True
        mxs = np.arange(1, M+1)
        for mx in range(1, M+1):
            mn = np.arange(0, mx + (1 if s==0 else 0))   # s=0: ax>=ay incl equal; s=1: ay>ax strictly -> mn<mx
            if len(mn)==0: continue
            ax = np.full(len(mn), mx) if s==0 else mn
            ay = mn if s==0 else np.full(len(mn), mx)
            dx = (-ax if xn else ax).astype(np.int16); dy = (-ay if yn else ay).astype(np.int16)
            q = O.orientation_bins(dx, dy).astype(int)
            # with xn and ax==0 (dx=-0 = 0) the sign test x<0 is false: handle separately below
            d = np.diff(q)
            if not (np.all(d>=0) or np.all(d<=0)): ok=False
            cls = np.abs(q - q[0])
            vals.add((int(q[0]), tuple(sorted(set(q.tolist())))))
            for k in (1,2):
                idx = np.nonzero(cls>=k)[0]
                if len(idx):
                    t = int(mn[idx[0]])
                    upper[k-1] = min(upper[k-1], Fraction(t, mx))
                    lower[k-1] = max(lower[k-1], Fraction(t-1, mx))
                else:
                    lower[k-1] = max(lower[k-1], Fraction(int(mn[-1]), mx))
        res[(s,xn,yn)] = (ok, lower, upper, vals)
        print((s,xn,yn), 'monotone', ok, 'thr1 in (%.7f, %.7f]' % (float(lower[0]), float(upper[0])), 'thr2 in (%.7f, %.7f]' % (float(lower[1]), float(upper[1])), 'feasible', lower[0]<upper[0], lower[1]<upper[1])
        qs = sorted(set(v for _,vs in vals for v in vs)); print('   q16 values', qs, ' q at mn=0:', sorted(set(a for a,_ in vals)))

def simplest_between(lo, hi):
    """simplest fraction f with lo < f <= hi (Stern-Brocot)"""
    a, b, c, d = 0, 1, 1, 0
    while True:
        m = Fraction(a + c, b + d)
        if m <= lo: a, b = m.numerator, m.denominator
        elif m > hi: c, d = m.numerator, m.denominator
        else: return m
lo1 = max(res[k][1][0] for k in res if k != (0,0,1)); hi1 = min(res[k][2][0] for k in res if k != (0,0,1))
lo2 = max(res[k][1][1] for k in res if k != (0,0,1)); hi2 = min(res[k][2][1] for k in res if k != (0,0,1))
print('thr1', lo1, hi1, float(lo1), float(hi1)); print('thr2', lo2, hi2, float(lo2), float(hi2))
f1 = simplest_between(lo1, hi1); f2 = simplest_between(lo2, hi2)
print('simplest', f1, float(f1), f2, float(f2))
# exhaustive check of the integer rule against the float pipeline
g = np.arange(-M, M+1, dtype=np.int16)
DX, DY = np.meshgrid(g, g)
q = O.orientation_bins(DX.ravel(), DY.ravel()).astype(np.int32)
ax, ay = np.abs(DX.ravel().astype(np.int64)), np.abs(DY.ravel().astype(np.int64))
mx, mn = np.maximum(ax, ay), np.minimum(ax, ay)
cls = ((mn * f1.denominator >= f1.numerator * mx) & (mx > 0)).astype(np.int64) + ((mn * f2.denominator >= f2.numerator * mx) & (mx > 0))
k = cls.copy()
s = ay > ax
k = np.where(s, 4 - k, k); k = np.where(DX.ravel() < 0, 8 - k, k); k = np.where(DY.ravel() < 0, 16 - k, k)
print('mismatches q16:', int((k != q).sum()), ' mismatches after &7:', int(((k & 7) != (q & 7)).sum()), 'of', len(q))
