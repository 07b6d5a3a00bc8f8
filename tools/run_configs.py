#!/usr/bin/env python3
"""Timing of the other BASELINE.json configurations on one MI355X (not the bench line: bench.py measures
configs[1]).  Stage-B inputs: seeded sparse one-hot maps + random templates, some planted, so
the template loop (matchClass) is isolated; the pyramid build is timed separately on a synthetic scene.
Prints one JSON line per configuration; --check compares the match multiset with the CPU oracle."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: rows, cols, n_templates, nf per level, template box, channels for the build timing
    "config1_case1_shape": (1024, 1024, 360, [128, 64], 260, 3),
    "config3_2048_3600x63": (2048, 2048, 3600, [63, 31], 260, 1),
    "config4_4096_8191feat_x360": (4096, 4096, 360, [8191, 4095], 1024, 1),
    "config5_1920x1072_1000": (1072, 1920, 1000, [128, 64], 260, 3),
}

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    import torch
    from shape_based_matching_amd import capi, synth
    from shape_based_matching_amd.templates import MATCH_DTYPE
    dev = torch.device("cuda", 0)
    T = (4, 8)
    for name, (rows, cols, nt, nf, box, ch) in CONFIGS.items():
        if args.only and args.only not in name:
            continue
        maps, ts = synth.stage_b(1234, rows, cols, T, nt, nf, templ_size=box, plant_every=40)
        ctx = capi.Context(T=T, max_candidates=1 << 22)
        ctx.upload_templates(ts)
        for l in range(2):
            ctx.set_quantized(l, maps[l])
        cap = 1 << 20
        d_out = torch.zeros(cap * MATCH_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        d_cnt = torch.zeros(2, dtype=torch.int32, device=dev)
        stream = torch.cuda.Stream(device=dev)
        def loop(n):
            with torch.cuda.stream(stream):
                for _ in range(n):
                    ctx.match_templates_device(90.0, d_out.data_ptr(), cap, d_cnt.data_ptr(), stream=stream.cuda_stream)
            stream.synchronize()
        loop(3)
        t0 = time.perf_counter(); loop(args.iters); t_loop = (time.perf_counter() - t0) / args.iters
        cnt = d_cnt.cpu().numpy()
        coarse_bytes = ctx.coarse_bytes()
        ctx.set_profiling(True); loop(1); tim = ctx.timings(); n_cand, refine_bytes = ctx.stats(); ctx.set_profiling(False)
        # pyramid build on a synthetic scene of the same size
        scene = synth.scene_bgr(7, rows, cols) if ch == 3 else synth.scene_gray(7, rows, cols)
        d_img = torch.from_numpy(scene).to(dev)
        ctx2 = capi.Context(T=T)
        ctx2.upload_templates(ts.subset(range(1)))
        def build(n):
            with torch.cuda.stream(stream):
                for _ in range(n):
                    ctx2.match_device(d_img.data_ptr(), rows, cols, cols * ch, ch, 99.0, d_out.data_ptr(), cap, d_cnt.data_ptr(), stream=stream.cuda_stream)
            stream.synchronize()
        build(3)
        t0 = time.perf_counter(); build(args.iters); t_build = (time.perf_counter() - t0) / args.iters
        out = {"config": name, "frame": [rows, cols, ch], "templates": nt, "features": nf,
               "template_loop_ms": t_loop * 1e3, "build_plus_1_template_ms": t_build * 1e3,
               "templates_Mpx_per_s_loop_only": nt * rows * cols / 1e6 / t_loop,
               "matches": int(cnt[0]), "overflow": int(cnt[1]), "coarse_candidates": n_cand,
               "coarse_bytes": coarse_bytes, "refine_bytes": refine_bytes,
               "kernels_us": {k: round(v * 1e3, 1) for k, v in tim},
               "coarse_algorithmic_GBps": coarse_bytes / max(dict(tim).get("k_similarity_coarse", 1e9) * 1e-3, 1e-12) / 1e9}
        if args.check:
            from oracle import oracle as O
            pyr = O.Pyramid.from_quantized(maps, T)
            want = pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, 90.0, n_threads=os.cpu_count() or 1)
            got = d_out.cpu().numpy().view(MATCH_DTYPE)[: cnt[0]]
            out["parity_with_oracle"] = sorted(got.tolist()) == sorted(want.tolist())
            t0 = time.perf_counter(); pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, 90.0, n_threads=1); out["oracle_loop_1thread_ms"] = (time.perf_counter() - t0) * 1e3
        print(json.dumps(out), flush=True)
        ctx.close(); ctx2.close()

if __name__ == "__main__":
    main()
