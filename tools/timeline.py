#!/usr/bin/env python3
"""Print the GPU timeline of the last steps from a rocprofv3 kernel-trace (+memory-copy) CSV."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '')[:40]) for r in rows]
if len(sys.argv) > 2 and sys.argv[2]:
    for r in csv.DictReader(open(sys.argv[2])):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', '')))
ev.sort()
n = int(sys.argv[3]) if len(sys.argv) > 3 else 30
sel = ev[-n:]
t0 = sel[0][0]
prev_end = None
for s, e, name in sel:
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{(s - t0) / 1e3:9.1f}us  dur {(e - s) / 1e3:7.1f}us  gap {gap:6.1f}us  {name}")
    prev_end = e
