#!/usr/bin/env python3
"""print the essentials of bench.py JSON lines: value, step time, per-kernel launch times, secondary frames"""
import json
import sys

for path in sys.argv[1:]:
    try:
        d = json.loads(open(path).read().strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001
        print(path, "ERR", e)
        continue
    c = d["config"]
    print(path, f"value {d['value'] / 1e6:.2f} M  {d['ms_per_step'] * 1e3:.1f} us/step  {c.get('us_per_frame', 0):.2f} us/frame  [{c.get('parallelism')}]")
    print("   kernels us:", {k: [round(x, 1) for x in v["launch_us"]] for k, v in d["kernels"].items()})
    extra = {k: round(v, 2) for k, v in c.items() if k.endswith("us_per_frame") and k != "us_per_frame"}
    if extra:
        print("   other frames us/frame:", extra)
    if "cpu_baseline" in d:
        b = d["cpu_baseline"]
        print("   cpu:", round(b["value"]), "on", b["cores"], "threads;", b.get("ms_per_match_by_threads"))
    print("   roofline:", d["roofline"]["kernel"], round(d["roofline"]["frac"], 4), "traffic", d["roofline"].get("traffic"))
