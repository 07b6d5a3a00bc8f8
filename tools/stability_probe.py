#!/usr/bin/env python3
"""Is a slow bench run slow for the whole process (placement of its allocations / streams) or only for some windows
(transient: clocks, other tenants)?  Times W consecutive windows of K steps of the bench's default workload in ONE
process, for a given frames-per-call x slots shape.  usage: python tools/stability_probe.py <batch> <inflight> [windows] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from shape_based_matching_amd import capi

B, S = int(sys.argv[1]), int(sys.argv[2])
W = int(sys.argv[3]) if len(sys.argv) > 3 else 10
K = int(sys.argv[4]) if len(sys.argv) > 4 else 100
dev = torch.device("cuda", 0)
ts, frame = bench.case1_templates(360), bench.case1_frame("case1", 1024, 1024)
d_img = torch.from_numpy(np.stack([np.roll(frame, 8 * b, axis=1) for b in range(B)])).to(dev)
cap = 256
slots = []
for i in range(S):
    c = capi.Context(T=bench.T_LEVELS, weak_threshold=30.0, device_id=0)
    c.upload_templates(ts)
    st = torch.cuda.Stream(device=dev)
    h = torch.zeros(16 + B * cap * 24, dtype=torch.uint8).pin_memory()
    d = torch.zeros(16 * B + B * cap * 24, dtype=torch.uint8, device=dev)
    c.set_result_mirror(h.data_ptr() + 16 * B if False else 0, 0)
    slots.append((c, st, d))
def step(k):
    c, st, d = slots[k % S]
    c.match_batch_device(d_img.data_ptr(), frame.size, B, 1024, 1024, 3072, 3, bench.THRESHOLD, d.data_ptr() + 16 * B, cap, d.data_ptr(),
                         stream=st.cuda_stream)
for k in range(40):
    step(k)
torch.cuda.synchronize()
res = []
for w in range(W):
    t0 = time.perf_counter()
    for k in range(K):
        step(k)
    torch.cuda.synchronize()
    res.append((time.perf_counter() - t0) / (K * B) * 1e6)
print(f"batch {B} slots {S}: us/frame per window:", " ".join(f"{x:.1f}" for x in res))
