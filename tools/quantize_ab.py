#!/usr/bin/env python3
"""A/B of the two gradient kernels (tile k_quantize vs row-streaming k_quantize_stream) on one MI355X:
per-launch kernel time of each pyramid level for several geometries, batch sizes and rows-per-wave, on a fully
textured frame and on a constant one.  One JSON line per case.  usage: python tools/quantize_ab.py [--quick]"""
import argparse, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--mixed", action="store_true", help="frames that are partly constant (balance of the work items)")
    args = ap.parse_args()
    import torch
    from shape_based_matching_amd import capi
    from shape_based_matching_amd.templates import MATCH_DTYPE, TemplateSet
    dev = torch.device("cuda", 0)
    ts = TemplateSet.load_npz(os.path.join(ROOT, "tests", "golden", "case1_templates.npz")).subset(range(2))
    img = np.load(os.path.join(ROOT, "tests", "golden", "case1_test_bgr.npz"))["bgr"]
    cases = [(1024, 1024, 3, 16), (1024, 1024, 1, 16), (1024, 1024, 3, 2), (1024, 1024, 3, 4), (1024, 1024, 3, 8), (1072, 1920, 3, 8), (2048, 2048, 1, 4), (512, 512, 3, 16), (512, 512, 3, 64)]
    if args.quick:
        cases = cases[:2]
    stream = torch.cuda.Stream(device=dev)
    for rows, cols, ch, B in cases:
        reps = (-(-rows // img.shape[0]), -(-cols // img.shape[1]), 1)
        tex = np.ascontiguousarray(np.tile(img, reps)[:rows, :cols])
        if ch == 1:
            tex = np.ascontiguousarray(tex[:, :, 1])
        half = tex.copy()
        half[rows // 2:] = 40
        emb = np.zeros_like(tex)
        ih, iw = min(img.shape[0], rows), min(img.shape[1], cols)
        sub = img[:ih, :iw] if ch == 3 else img[:ih, :iw, 1]
        emb[(rows - ih) // 2:(rows - ih) // 2 + ih, (cols - iw) // 2:(cols - iw) // 2 + iw] = sub
        kinds = (("textured", tex), ("constant", np.full_like(tex, 40))) if not args.mixed else (("half", half), ("embedded", emb))
        for kind, frame in kinds:
            d_img = torch.from_numpy(np.stack([frame] * B)).to(dev)
            cap = 4096
            d_out = torch.zeros(B * cap * MATCH_DTYPE.itemsize, dtype=torch.uint8, device=dev)
            d_cnt = torch.zeros(B * 2, dtype=torch.int32, device=dev)
            modes = [("tile", 0)] + [("stream", h) for h in (0, 8, 12, 16, 20, 24, 28, 32, 40)]
            ctx = capi.Context(T=(4, 8), max_candidates=1 << 16)
            ctx.upload_templates(ts)
            for mode, hs in modes:
                ctx.set_quantize_mode(mode, hs)

                def run(n):
                    for _ in range(n):
                        ctx.match_batch_device(d_img.data_ptr(), frame.size, B, rows, cols, cols * ch, ch, 99.9, d_out.data_ptr(), cap,
                                               d_cnt.data_ptr(), stream=stream.cuda_stream)
                    stream.synchronize()
                run(3)
                ctx.set_profiling(True, accumulate=True)
                n = 20
                run(n)
                t = [ms for name, ms in ctx.timings() if name == "k_quantize"]
                ctx.set_profiling(False)
                a = np.asarray(t).reshape(n, -1).mean(axis=0) * 1e3
                print(json.dumps({"frame": [rows, cols, ch], "batch": B, "content": kind, "mode": mode, "rows_per_wave": hs,
                                  "level_us": [round(float(x), 1) for x in a]}), flush=True)
            ctx.close()


if __name__ == "__main__":
    main()
