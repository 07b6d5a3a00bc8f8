#!/usr/bin/env python3
"""PCIe-inclusive rates of the bench workload (never bench.py's `value`): (1) the host entry point sbm_match (frame in
pageable host memory, synchronous); (2) batches of frames streamed from pinned host memory on a copy stream while the
previous batch is matched (two device buffers, two slots)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from shape_based_matching_amd import capi
from shape_based_matching_amd.templates import MATCH_DTYPE

ts, frame = bench.case1_templates(360), bench.case1_frame("case1", 1024, 1024)
dev = torch.device("cuda", 0)
R, C = frame.shape[:2]
ctx = capi.Context(T=bench.T_LEVELS, weak_threshold=30.0, device_id=0)
ctx.upload_templates(ts)
for _ in range(5):
    ctx.match(frame, bench.THRESHOLD)
t0 = time.perf_counter()
n = 50
for _ in range(n):
    recs = ctx.match(frame, bench.THRESHOLD)
t1 = time.perf_counter()
print(f"sbm_match (pageable host frame in, host list out, synchronous): {(t1 - t0) / n * 1e6:.1f} us per frame, {len(recs)} matches")

B = 16
cap = 256
fb = R * C * 3
h_frames = torch.from_numpy(np.stack([np.roll(frame, 8 * b, axis=1) for b in range(B)])).pin_memory()
slots = []
for i in range(2):
    c = capi.Context(T=bench.T_LEVELS, weak_threshold=30.0, device_id=0)
    c.upload_templates(ts)
    slots.append(dict(ctx=c, d_img=torch.empty((B, R, C, 3), dtype=torch.uint8, device=dev), comp=torch.cuda.Stream(device=dev),
                      copy=torch.cuda.Stream(device=dev), d_out=torch.zeros(B * cap * 24, dtype=torch.uint8, device=dev),
                      d_cnt=torch.zeros(2 * B, dtype=torch.int32, device=dev), ev_copy=torch.cuda.Event(), ev_done=torch.cuda.Event()))
def step(k):
    s = slots[k % 2]
    with torch.cuda.stream(s["copy"]):
        s["copy"].wait_event(s["ev_done"])          # the previous batch in this buffer has been matched
        s["d_img"].copy_(h_frames, non_blocking=True)
        s["ev_copy"].record(s["copy"])
    s["comp"].wait_event(s["ev_copy"])
    s["ctx"].match_batch_device(s["d_img"].data_ptr(), fb, B, R, C, C * 3, 3, bench.THRESHOLD, s["d_out"].data_ptr(), cap,
                                s["d_cnt"].data_ptr(), stream=s["comp"].cuda_stream)
    s["ev_done"].record(s["comp"])
for s in slots:
    s["ev_done"].record(s["comp"])
for k in range(10):
    step(k)
torch.cuda.synchronize()
n = 100
t0 = time.perf_counter()
for k in range(n):
    step(k)
torch.cuda.synchronize()
t1 = time.perf_counter()
per = (t1 - t0) / (n * B)
print(f"pinned host -> HBM copy overlapped with the match of the previous batch ({B} frames per batch): {per * 1e6:.1f} us per frame "
      f"= {fb / per / 1e9:.1f} GB/s over PCIe, counts {slots[0]['d_cnt'].cpu().numpy()[:4].tolist()}")
