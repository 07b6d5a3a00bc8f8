#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection CSV per (kernel, grid)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(dict)
for r in rows:
    name = r['Kernel_Name'].split('(')[0].replace('void ', '')
    key = (name, r['Grid_Size'])
    agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
    dur[key][r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for key, d in sorted(agg.items()):
    if len(sys.argv) > 2 and sys.argv[2] not in key[0]:
        continue
    m = {k: sum(v) / len(v) for k, v in d.items()}
    us = sum(dur[key].values()) / len(dur[key])
    w = max(m.get('SQ_WAVES', 1), 1)
    line = f"{key[0][:30]:30s} grid {key[1]:>8s} n={len(dur[key]):3d} dur {us:6.1f}us waves {w:6.0f}"
    for c in sorted(m):
        if c in ('SQ_WAVES',):
            continue
        if c.startswith('SQ_INSTS') :
            line += f" {c[9:]}/w {m[c] / w:7.0f}"
        elif c in ('SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES'):
            line += f" {c[3:]} {m[c]:9.0f}"
        elif c.startswith('SQ_'):
            line += f" {c[3:]}% {100 * m[c] / max(m.get('SQ_WAVE_CYCLES', 1), 1):4.0f}"
        else:
            line += f" {c} {m[c]:.0f}"
    print(line)
