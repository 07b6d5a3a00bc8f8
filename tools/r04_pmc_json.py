#!/usr/bin/env python3
"""rocprofv3 --pmc passes of one bench configuration -> one JSON entry (profiles/r04_pmc.json, read by bench.py).

usage: r04_pmc_json.py <out.json> <config> <fetch.csv> <write.csv> <sq_a.csv> <sq_b.csv> <tcc.csv>
Per kernel of the step (the (kernel, grid) group with the most launches; k_quantize: mean over the levels' launches):
  hbm_bytes_per_launch   (FETCH_SIZE [x2 for the kernels whose global reads are 16 B per lane: MI355X_MICROARCH.md, HBM] +
                          WRITE_SIZE) * 1024; the two counters come from separate passes, as the guide prescribes
  waves, valu_per_wave, salu_per_wave, vmem_rd_per_wave (SQ_WAVES, SQ_INSTS_*), wait / active percentages of the wave time
  l2_requests (TCC_HIT_sum + TCC_MISS_sum; a request is one 128-byte line), l2_hit_rate
"""
import collections
import csv
import json
import os
import sys

ALIAS = {"k_build_lm_rows": "k_build_lm", "k_quantize_stream": "k_quantize", "k_similarity_coarse_bits": "k_similarity_coarse",
         "k_similarity_coarse_wave": "k_similarity_coarse", "k_pack_bitplanes": "k_pack_bitplanes", "k_similarity_local": "k_similarity_local", "k_similarity_local_bits": "k_similarity_local"}
WIDE_READS = {"k_build_lm_rows"}  # 16 B per lane streaming reads: FETCH_SIZE counts half of them on gfx950


def groups(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(list)))
    dur = collections.defaultdict(lambda: collections.defaultdict(dict))
    for r in csv.DictReader(open(path)):
        full = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sbm::", "")
        name = full.split("<")[0]
        key = (full, r.get("Grid_Size", ""))
        acc[name][key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[name][key][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    out = {}
    for name, g in acc.items():
        most = max(len(next(iter(c.values()))) for c in g.values())
        keep = [k for k, c in g.items() if len(next(iter(c.values()))) * 2 >= most]  # the step's launches (all levels)
        m = collections.defaultdict(list)
        d = []
        for k in keep:
            for cname, v in g[k].items():
                m[cname] += v
            d += list(dur[name][k].values())
        out[name] = ({c: sum(v) / len(v) for c, v in m.items()}, sum(d) / len(d), len(keep))
    return out


def main():
    out_path, config, fetch, write, sqa, sqb, tcc = sys.argv[1:8]
    data = json.load(open(out_path)) if os.path.exists(out_path) else {
        "_source": "rocprofv3 --pmc: FETCH_SIZE, WRITE_SIZE, SQ passes, each in a run of its own (tools/r04_profile.sh)",
        "_units": "bytes / instructions per launch; FETCH_SIZE and WRITE_SIZE are KiB; x2 on FETCH_SIZE only for 16 B-per-lane streaming reads"}
    F, W, A, B, C = groups(fetch), groups(write), groups(sqa), groups(sqb), groups(tcc)
    entry = {}
    for name in sorted(set(F) | set(W) | set(A) | set(B)):
        if name not in ALIAS:
            continue
        f = F.get(name, ({}, 0, 0))[0].get("FETCH_SIZE", 0.0)
        w = W.get(name, ({}, 0, 0))[0].get("WRITE_SIZE", 0.0)
        a, dur_a, n_kinds = A.get(name, ({}, 0.0, 0))
        b = B.get(name, ({}, 0.0, 0))[0]
        waves = max(a.get("SQ_WAVES", b.get("SQ_WAVES", 1.0)), 1.0)
        wc = max(a.get("SQ_WAVE_CYCLES", 1.0), 1.0)
        e = {"hbm_bytes_per_launch": (f * (2.0 if name in WIDE_READS else 1.0) + w) * 1024.0,
             "raw_KiB": {"FETCH_SIZE": f, "WRITE_SIZE": w, "fetch_x2": name in WIDE_READS},
             "waves": waves, "launch_us_under_counters": dur_a,
             "valu_per_wave": b.get("SQ_INSTS_VALU", 0.0) / waves, "salu_per_wave": b.get("SQ_INSTS_SALU", 0.0) / waves,
             "vmem_rd_per_wave": b.get("SQ_INSTS_VMEM_RD", 0.0) / waves, "lds_per_wave": b.get("SQ_INSTS_LDS", 0.0) / waves,
             "active_valu_pct": 100.0 * a.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, "wait_inst_any_pct": 100.0 * a.get("SQ_WAIT_INST_ANY", 0.0) / wc,
             "wait_any_pct": 100.0 * a.get("SQ_WAIT_ANY", 0.0) / wc, "wave_cycles": wc}
        t = C.get(name, ({}, 0, 0))[0]
        req = t.get("TCC_HIT_sum", 0.0) + t.get("TCC_MISS_sum", 0.0)
        e["l2_requests"] = req
        e["l2_hit_rate"] = t.get("TCC_HIT_sum", 0.0) / req if req else None
        entry[ALIAS[name]] = e
    data[config] = entry
    json.dump(data, open(out_path, "w"), indent=1)
    for k, e in entry.items():
        print(config, k, {x: (round(y, 1) if isinstance(y, float) else y) for x, y in e.items() if x != "raw_KiB"})


if __name__ == "__main__":
    main()
