"""Does the coarse pass's exact pruning bite?  Times the template loop of a config-4-like workload over thresholds
(the prefix k1 shrinks as the threshold rises) on the GPU box.  usage: python tools/prune_probe.py [c3|c4] [n_templates]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shape_based_matching_amd import capi, synth  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
plant = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows, nf, box = (2048, [63, 31], 260) if cfg == "c3" else (4096, [8191, 4095], 1024)
maps, ts = synth.stage_b(1234, rows, rows, (4, 8), n, nf, templ_size=box, plant_every=plant)
print("density L1 %.4f" % (np.count_nonzero(maps[1]) / maps[1].size), flush=True)
torch.cuda.init()
ctx = capi.Context(T=(4, 8), weak_threshold=30.0, device_id=0, max_candidates=1 << 22)
ctx.upload_templates(ts)
for l in range(2):
    ctx.set_quantized(l, maps[l])
cap = 1 << 16
out = torch.zeros(16 + cap * 16, dtype=torch.uint8, device="cuda")
s = torch.cuda.Stream()
for thr in (50.0, 70.0, 80.0, 90.0, 95.0, 99.0):
    for _ in range(2):
        ctx.match_templates_device(thr, out.data_ptr() + 16, cap, out.data_ptr(), stream=s.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        ctx.match_templates_device(thr, out.data_ptr() + 16, cap, out.data_ptr(), stream=s.cuda_stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    cnt = out[:8].cpu().numpy().view(np.int32)
    print(f"thr {thr}: {dt * 1e3:.3f} ms  matches {cnt[0]} overflow {cnt[1]}", flush=True)
