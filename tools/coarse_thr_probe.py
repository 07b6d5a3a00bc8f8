"""Coarse-pass launch time of the bench's 16-frame case1 step over thresholds (the pruning prefix k1 shrinks as the
threshold rises: 90 -> 20 features, 99 -> 8): separates the per-item overhead from the feature loads.  GPU box.
usage: python tools/coarse_thr_probe.py [case1|tiled]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from shape_based_matching_amd import capi  # noqa: E402
from shape_based_matching_amd.templates import MATCH_DTYPE  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "case1"
torch.cuda.init()
ts = bench.case1_templates(360)
frame = bench.case1_frame(kind, 1024, 1024)
B = 16
frames = np.stack([np.roll(frame, 8 * b, axis=1) for b in range(B)])
d_img = torch.from_numpy(frames).cuda()
cap = 4096
d_out = torch.zeros(B * cap * MATCH_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
d_cnt = torch.zeros(B * 2, dtype=torch.int32, device="cuda")
s = torch.cuda.Stream()
for mode in ("wave", "block"):
    ctx = capi.Context(T=(4, 8), weak_threshold=30.0, device_id=0)
    ctx.set_coarse_mode(mode)
    ctx.upload_templates(ts)
    for thr in (60.0, 80.0, 85.0, 90.0, 95.0, 99.0, 100.0):
        def run():
            ctx.match_batch_device(d_img.data_ptr(), frames[0].size, B, 1024, 1024, 3072, 3, thr, d_out.data_ptr(), cap, d_cnt.data_ptr(),
                                   stream=s.cuda_stream)
        for _ in range(5):
            run()
        s.synchronize()
        ctx.set_profiling(True, accumulate=True)
        n = 50
        for _ in range(n):
            run()
        s.synchronize()
        t = {}
        for name, ms in ctx.timings():
            t[name] = t.get(name, 0.0) + ms
        ctx.set_profiling(False)
        cnt = d_cnt.cpu().numpy().reshape(B, 2)
        print(f"{mode} thr {thr}: coarse {t.get('k_similarity_coarse', 0) / n * 1e3:.1f} us  local {t.get('k_similarity_local', 0) / n * 1e3:.1f} us"
              f"  matches/frame {cnt[:, 0].mean():.0f}", flush=True)
    ctx.close()
