#!/usr/bin/env python3
"""Host-side cost of one sbm_match_device enqueue vs. GPU time per step (bench workload)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from shape_based_matching_amd import capi
ts, frame = bench.load_workload(1)
if os.environ.get("TINY"):  # minimal GPU work: 8 templates, black frame -> every kernel sits on the launch floor
    ts = ts.subset(range(8)); frame = frame * 0
dev = torch.device("cuda", 0)
d_img = torch.from_numpy(frame).to(dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
slots = []
for i in range(n):
    ctx = capi.Context(T=bench.T_LEVELS, weak_threshold=30.0, device_id=0)
    ctx.upload_templates(ts)
    st = torch.cuda.Stream(device=dev)
    buf = torch.zeros(16 + 1024 * 24, dtype=torch.uint8, device=dev)
    slots.append((ctx, st, buf))
def run(k):
    ctx, st, buf = slots[k % n]
    ctx.match_device(d_img.data_ptr(), 1024, 1024, 3072, 3, 90.0, buf.data_ptr() + 16, 1024, buf.data_ptr(), stream=st.cuda_stream)
for k in range(100): run(k)
torch.cuda.synchronize()
for steps in (50, 400):
    t0 = time.perf_counter()
    for k in range(steps): run(k)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"inflight {n} steps {steps}: enqueue {1e6*(t1-t0)/steps:.2f} us/step, total {1e6*(t2-t0)/steps:.2f} us/step")
# two host threads, each enqueuing on its own slots
import threading
if n >= 2:
    def worker(tid, steps, nthreads):
        for k in range(steps):
            run(tid + nthreads * (k % (n // nthreads)))
    for nthreads in (2, 4):
        if n % nthreads: continue
        steps = 400
        th = [threading.Thread(target=worker, args=(i, steps, nthreads)) for i in range(nthreads)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"inflight {n}, {nthreads} host threads: total {1e6*(t2-t0)/(steps*nthreads):.2f} us/step")
