#!/usr/bin/env python3
"""Randomised end-to-end check of the batch entry point against the CPU oracle (GPU box; test infrastructure).
Random frame geometry (multiples of 16), pyramid, channel count, batch size, template subset, threshold, mask, gradient-kernel
mode, coarse-pass kernel, refinement order and entry point; every frame's match multiset must equal the oracle's.
usage: python tools/fuzz_match.py [n_cases] [seed]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from shape_based_matching_amd import capi, synth  # noqa: E402
from shape_based_matching_amd.templates import MATCH_DTYPE, TemplateSet, from_pyramids  # noqa: E402


def multiset(recs):
    return sorted(np.ascontiguousarray(recs, MATCH_DTYPE).tolist())


def run(n_cases, seed, verbose=True):
    rs = np.random.RandomState(seed)
    O.build()
    O.lib()
    torch.cuda.init()
    dev = torch.device("cuda", 0)
    all_ts = TemplateSet.load_npz(os.path.join(ROOT, "tests", "golden", "case1_templates.npz"))
    img = np.load(os.path.join(ROOT, "tests", "golden", "case1_test_bgr.npz"))["bgr"]
    t0 = time.time()
    n_matches = 0
    for case in range(n_cases):
        rows = 16 * int(rs.randint(30, 70))
        cols = 64 * int(rs.randint(7, 18))  # the batch entry point wants (cols / T) % 4 == 0 at every level
        # round 3: the pyramid is drawn too (the batch entry point takes T in {4, 8}); three levels need the geometry to
        # stay 4-cell aligned one level further down
        T = [(4, 8), (4, 8), (4, 8), (4, 8), (4,), (8,), (8, 8), (4, 4), (4, 8, 8)][int(rs.randint(0, 9))]
        if len(T) == 3:
            rows, cols = (rows + 31) // 32 * 32, (cols + 127) // 128 * 128
        ch = int(rs.choice([1, 3]))
        B = int(rs.randint(1, 10))
        n_t = int(rs.choice([1, 3, 7, 30, 90, 200]))
        idx = sorted(rs.choice(all_ts.n_templates, n_t, replace=False).tolist())
        ts = all_ts.subset(idx)
        if T != (4, 8):  # the fixture's 2-level templates re-cut to the pyramid: level l from fixture level min(l, 1), halved below
            pyrs = []
            for t in idx:
                lv = []
                for l in range(len(T)):
                    src = all_ts.levels[t, min(l, 1)]
                    f = all_ts.features[src["feature_offset"]: src["feature_offset"] + src["n_features"]]
                    scale = 1 if l < 2 else 2
                    feats = np.stack([f["x"] // scale, f["y"] // scale, f["label"]], axis=1)
                    lv.append({"width": int(src["width"]) // scale, "height": int(src["height"]) // scale, "tl_x": 0, "tl_y": 0,
                               "pyramid_level": l, "features": feats})
                pyrs.append(lv)
            ts = from_pyramids(pyrs, "t")
        thr = float(rs.choice([55.0, 70.0, 80.0, 88.0, 93.0, 98.0]))
        kind = rs.choice(["embed", "tile", "scene"])
        frames = []
        for b in range(B):
            if kind == "scene":
                fr = synth.scene_bgr(int(rs.randint(1 << 20)), rows, cols)
            elif kind == "tile":
                reps = (-(-rows // img.shape[0]), -(-cols // img.shape[1]), 1)
                fr = np.roll(np.tile(img, reps)[:rows, :cols], int(rs.randint(0, 64)), axis=1)
            else:
                h, w = min(rows, img.shape[0]), min(cols, img.shape[1])
                fr = synth.embed(img[:h, :w], rows, cols, int(rs.randint(0, rows - h + 1)), int(rs.randint(0, cols - w + 1)))
            frames.append(np.ascontiguousarray(fr if ch == 3 else fr[:, :, 1]))
        mask = None
        if rs.randint(0, 4) == 0:
            mask = np.zeros((rows, cols), np.uint8)
            mask[rows // 8: rows - rows // 6, cols // 7: cols - cols // 9] = 255
        cmode = str(rs.choice(["", "block", "wave", "bits", "bytes"]))
        qmode = str(rs.choice(["auto", "tile", "stream"]))
        hs = int(rs.choice([0, 6, 8, 16, 28]))
        ctx = capi.Context(T=T, weak_threshold=30.0, device_id=0)
        ctx.upload_templates(ts)
        ctx.set_quantize_mode(qmode, hs)
        ctx.set_coarse_mode(cmode)
        ctx.set_refine_order(str(rs.choice(["auto", "slots", "list"])))
        depth = int(rs.choice([1, 3]))  # round 3: throughput sizing of the launches (identical results)
        ctx.set_pipeline_depth(depth)
        # round 3: the entry point is drawn too -- the device batch, the pipelined host batch (sub-batches of 1..4), or the
        # build-sharded step run on one GPU (every row band launched here); bands need rows_l % n == 0 with an even quotient
        entry = str(rs.choice(["device", "host", "banded"]))
        n_bands = 0
        if entry == "banded":
            ok = [n for n in (2, 3, 4, 5, 6, 8) if (rows // 2) % n == 0 and ((rows // 2) // n) % 2 == 0 and rows % n == 0 and (rows // n) % 2 == 0]
            if qmode == "tile" or not ok or T != (4, 8):
                entry = "device"
            else:
                n_bands = int(rs.choice(ok))
        cap, rec = 8192, MATCH_DTYPE.itemsize
        stream = torch.cuda.Stream(device=dev)
        if entry == "host":
            lists = ctx.match_batch_host(frames, thr, cap=cap, sub_batch=int(rs.randint(1, 5)), mask=mask, split=bool(rs.randint(0, 2)))
            cnt = np.array([[len(l), 0] for l in lists], np.int32)
            out = np.zeros((B, cap * rec), np.uint8)
            for f, l in enumerate(lists):
                out[f, : len(l) * rec] = np.ascontiguousarray(l).view(np.uint8)
        else:
            d_img = torch.from_numpy(np.stack(frames)).to(dev)
            d_mask = torch.from_numpy(mask).to(dev) if mask is not None else None
            hdr = (8 * B + 15) // 16 * 16
            d_buf = torch.zeros(hdr + B * cap * rec, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            if entry == "banded":
                ctx.match_batch_device_banded(d_img.data_ptr(), frames[0].size, B, rows, cols, cols * ch, ch, thr, d_buf.data_ptr(), cap,
                                              n_bands=n_bands, stream=stream.cuda_stream, d_mask=d_mask.data_ptr() if d_mask is not None else 0)
            else:
                ctx.match_batch_device(d_img.data_ptr(), frames[0].size, B, rows, cols, cols * ch, ch, thr, d_buf.data_ptr() + hdr, cap,
                                       d_buf.data_ptr(), d_mask=d_mask.data_ptr() if d_mask is not None else 0, stream=stream.cuda_stream)
            stream.synchronize()
            h = d_buf.cpu().numpy()
            cnt = h[: 8 * B].view(np.int32).reshape(B, 2)
            out = h[hdr:].reshape(B, cap * rec)
        desc = (case, T, rows, cols, ch, B, n_t, thr, kind, mask is not None, cmode, qmode, hs, depth, entry, n_bands)
        for f in range(B):
            pyr = O.Pyramid.build(frames[f], list(T), 30.0, mask=mask)
            want = pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, thr, n_threads=min(16, os.cpu_count() or 1))
            pyr.free()
            if len(want) > cap:
                continue
            assert cnt[f, 1] == 0 and cnt[f, 0] == len(want), (desc, f, cnt[f].tolist(), len(want))
            assert multiset(out[f].view(MATCH_DTYPE)[: cnt[f, 0]]) == multiset(want), (desc, f)
            n_matches += len(want)
        ctx.close()
        if verbose:
            print("ok", desc, flush=True)
    if verbose:
        print(f"{n_cases} cases, {n_matches} matches compared, {time.time() - t0:.0f} s", flush=True)
    return n_matches


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
