#!/usr/bin/env python3
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection CSVs (separate passes, as the
MI355X guide prescribes) into profiles/pmc_traffic.json: HBM-side bytes per launch, per kernel.
FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reads
half the bytes of wide (16 B/lane) coalesced streaming reads; it is applied only to the kernels whose
global reads are 16 B/lane (k_build_lm_rows, k_similarity_coarse); the others use <= 4 B/lane loads,
for which the raw counter matched the known byte counts of this workload (k_quantize: 3.00 MiB read
for a 3 MiB frame)."""
import collections, csv, json, sys

def per_kernel(path, counter):
    """mean per launch over the launches of the timed steps: bench.py also issues a few single-frame launches
    (verification), so per kernel only the work-group count that occurs most often is kept, and for a kernel with
    several launches per step (k_quantize: one per level) the mean is taken over those"""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        full = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('sbm::', '')
        name = full.split('<')[0]
        acc[name][(full, r.get('Grid_Size', ''))].append(float(r['Counter_Value']))
    out = {}
    for name, groups in acc.items():
        most = max(len(v) for v in groups.values())
        vals = [x for v in groups.values() if len(v) * 2 >= most for x in v]  # the step's launches (all levels)
        out[name] = sum(vals) / len(vals)
    return out

fetch = per_kernel(sys.argv[1], 'FETCH_SIZE')
write = per_kernel(sys.argv[2], 'WRITE_SIZE')
wide = {'k_build_lm_rows', 'k_similarity_coarse', 'k_similarity_coarse_wave'}
out = {'_unit': 'bytes per launch (average over the launches of one step)', '_source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes', '_raw_KiB': {}}
# the step's gradient launches are the row-streaming kernel; the 16 x 64 tile kernel only serves the single-frame
# verification calls of bench.py and keeps its own key
alias = {'k_build_lm_rows': 'k_build_lm', 'k_quantize_stream': 'k_quantize', 'k_quantize': 'k_quantize_tile'} if 'k_quantize_stream' in (set(fetch) | set(write)) else {'k_build_lm_rows': 'k_build_lm'}
# likewise the batch's coarse pass is the one-wave-per-item kernel; the four-wave kernel serves single frames
if 'k_similarity_coarse_wave' in (set(fetch) | set(write)):
    alias.update({'k_similarity_coarse_wave': 'k_similarity_coarse', 'k_similarity_coarse': 'k_similarity_coarse_block'})
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, 0.0), write.get(k, 0.0)
    fc = f * (2.0 if k in wide else 1.0)
    out[alias.get(k, k)] = (fc + w) * 1024.0
    out['_raw_KiB'][alias.get(k, k)] = {'FETCH_SIZE': f, 'WRITE_SIZE': w, 'fetch_x2_correction': k in wide}
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print(json.dumps(out, indent=1))
