#!/usr/bin/env python3
"""bench.py — throughput of the LINE-2D match() hot path on MI355X.

A step = one pass of the hot path over one batch of frames that are already
resident in HBM: for every frame one whole Detector::match
(line2Dup.cpp:1078-1150) — gradient quantisation -> pyramid -> spread/response/
linearize -> similarity over this rank's template shard -> 16x16 refinement ->
match records — with the per-frame match lists all-gathered over RCCL when
N > 1 and stored in pinned host memory.  Metric: templates * Mpixels / s
(BASELINE.json), whole job: templates x Mpixels x frames per step x steps / time.

Workload (BASELINE.json configs[1], "case1 on 1x MI355X"): the reference's
case1 test image centred on a 1024 x 1024 BGR canvas, 360 case1 rotation
templates (131 / 71 features) per GPU, pyramid {4, 8}, threshold 90.  With N
GPUs the template set is N x 360 (weak scaling), sharded by contiguous ranges.
--batch frames per step (default 16: sbm_match_batch_device launches every kernel
once for the whole batch; frame b is the workload frame shifted 8*b columns) and
--inflight independent slots (contexts + streams, default 2) used round-robin;
--batch 1 --inflight 1 is strictly one frame at a time (sbm_match_device).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s
ROWS = COLS = 1024
N_TEMPLATES = 360
THRESHOLD = 90.0
T_LEVELS = (4, 8)
PREFETCH = 256  # capacity (records) of the per-frame match list exchanged between ranks / sent to the host


def load_workload(world: int, frame_kind: str = "case1"):
    from shape_based_matching_amd import synth
    from shape_based_matching_amd.templates import TemplateSet

    golden = os.path.join(ROOT, "tests", "golden")
    base = TemplateSet.load_npz(os.path.join(golden, "case1_templates.npz")).subset(range(N_TEMPLATES))
    shards = []
    for r in range(world):
        s = base.subset(range(N_TEMPLATES))
        s.class_ids = [f"test{r}"]
        shards.append(s)
    ts = TemplateSet.concat(shards)
    img = np.load(os.path.join(golden, "case1_test_bgr.npz"))["bgr"]
    if frame_kind == "tiled":
        # no constant region anywhere: the test image repeated over the whole canvas (robustness figure,
        # none of k_quantize's flat-tile shortcuts fire)
        reps = (-(-ROWS // img.shape[0]), -(-COLS // img.shape[1]), 1)
        frame = np.ascontiguousarray(np.tile(img, reps)[:ROWS, :COLS])
    else:
        frame = synth.embed(img, ROWS, COLS, (ROWS - img.shape[0]) // 2, (COLS - img.shape[1]) // 2)
    return ts, frame


def cpu_baseline(ts, frame, budget_s: float = 12.0):
    """The CPU oracle (a port of the reference algorithm, OpenMP over templates like
    line2Dup.cpp:1166-1170) timed on this host on the same frame and templates."""
    from oracle import oracle as O

    ncpu = os.cpu_count() or 1

    def run(threads, reps):
        t0 = time.perf_counter()
        n = 0
        for _ in range(reps):
            pyr = O.Pyramid.build(frame, list(T_LEVELS), 30.0)
            recs = pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, THRESHOLD, n_threads=threads)
            n = len(recs)
            pyr.free()
        return (time.perf_counter() - t0) / reps, n

    t1, n1 = run(1, 2)
    tn, _ = run(ncpu, 2) if ncpu > 1 else (t1, n1)
    threads = 1 if t1 <= tn else ncpu
    per = min(t1, tn)
    reps = max(3, int(budget_s / per))
    per, n = run(threads, reps)
    value = ts.n_templates * (frame.shape[0] * frame.shape[1] / 1e6) / per
    return {
        "value": value,
        "unit": "templates*Mpixels/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{reps} full match() calls of the bench frame with {ts.n_templates} templates "
                  f"({per * 1e3:.1f} ms each, {n} raw matches); host has {ncpu} logical cores",
        "ms_per_match": per * 1e3,
        # both ways of running the reference's OpenMP template loop (always report both)
        "ms_per_match_1_thread": t1 * 1e3,
        "ms_per_match_all_threads": tn * 1e3,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    ap.add_argument("--force-collective", action="store_true",
                    help="rehearsal: run the N>1 code path (RCCL all-gathers + host copy) with whatever world size")
    ap.add_argument("--inflight", type=int, default=2,
                    help="frames in flight per GPU: independent engine contexts + HIP streams used round-robin "
                         "(1 = strictly one frame at a time; the single-stream figure is always reported too)")
    ap.add_argument("--batch", type=int, default=16,
                    help="frames per step: a step is one sbm_match_batch_device call over this many frames (distinct "
                         "horizontal shifts of the workload frame); 1 = one sbm_match_device call per step")
    ap.add_argument("--frame", choices=("case1", "tiled"), default="case1",
                    help="case1: the reference's test image centred on a black canvas (BASELINE configs[1]); "
                         "tiled: the same image repeated over the whole canvas (no constant regions)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from shape_based_matching_amd import capi, sharding
    from shape_based_matching_amd.templates import MATCH_DTYPE

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    collective = world > 1 or args.force_collective
    if collective:
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"] = "127.0.0.1"
            os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    ts, frame = load_workload(world, args.frame)
    first, count = sharding.partition(sharding.coarse_work(ts, ROWS, COLS, T_LEVELS), world)[rank]

    cap = PREFETCH
    B = max(1, args.batch)
    # frame b of a batch = the workload frame rolled b * 8 columns (same content, same work, different bytes)
    batch_frames = np.stack([np.roll(frame, 8 * b, axis=1) for b in range(B)])
    d_img = torch.from_numpy(batch_frames).to(dev)
    FRAME_BYTES = ROWS * COLS * 3

    REC = MATCH_DTYPE.itemsize
    # per rank: B {n_matches, overflow} int32 pairs (padded to 16 bytes) in front of B blocks of cap records:
    # one buffer, one collective, one copy
    HDR = (8 * B + 15) // 16 * 16
    BUF = HDR + B * cap * REC

    class Slot:
        """one frame in flight: its own engine context (device buffers), stream and result buffers"""

        def __init__(self, main_stream):
            self.ctx = capi.Context(T=T_LEVELS, weak_threshold=30.0, device_id=local_rank)
            self.ctx.upload_templates(ts)
            self.ctx.select_range(first, count)
            if os.environ.get("SBM_GRAPH"):
                self.ctx.set_graph_mode(True)
            # always an explicit stream: handle 0 would mean "the context's own stream" to the C ABI and
            # the result copies below must be ordered after the kernels
            self.stream = torch.cuda.Stream(device=dev)
            self.d_buf = torch.zeros(BUF, dtype=torch.uint8, device=dev)       # this rank: header + records
            self.g_buf = torch.zeros(world * BUF, dtype=torch.uint8, device=dev)  # all ranks, gathered
            self.h_buf = torch.zeros(world * BUF, dtype=torch.uint8).pin_memory()
            if not collective:
                # single GPU: the last kernel stores the match list straight into pinned host memory
                self.ctx.set_result_mirror(self.h_buf.data_ptr() + HDR, self.h_buf.data_ptr())
            else:
                # one RCCL communicator per frame slot, bootstrapped over torch.distributed
                uid = torch.zeros(128, dtype=torch.uint8, device=dev)
                if rank == 0:
                    uid.copy_(torch.frombuffer(bytearray(capi.Context.comm_unique_id()), dtype=torch.uint8))
                dist.broadcast(uid, src=0)
                self.ctx.comm_init(world, rank, bytes(uid.cpu().numpy().tobytes()))

        def run(self):
            if collective and B > 1:
                self.ctx.match_batch_device_sharded(d_img.data_ptr(), FRAME_BYTES, B, ROWS, COLS, COLS * 3, 3, THRESHOLD,
                                                    self.d_buf.data_ptr(), cap, self.g_buf.data_ptr(),
                                                    gathered_mirror=self.h_buf.data_ptr(), stream=self.stream.cuda_stream)
            elif collective:
                # match of this rank's template shard + the exchange step (ncclAllGather over xGMI, issued by the
                # library on the same stream) + copy of the gathered lists into pinned host memory
                self.ctx.match_device_sharded(d_img.data_ptr(), ROWS, COLS, COLS * 3, 3, THRESHOLD, self.d_buf.data_ptr(), cap,
                                              self.g_buf.data_ptr(), gathered_mirror=self.h_buf.data_ptr(),
                                              stream=self.stream.cuda_stream)
            elif B > 1:
                self.ctx.match_batch_device(d_img.data_ptr(), FRAME_BYTES, B, ROWS, COLS, COLS * 3, 3, THRESHOLD,
                                            self.d_buf.data_ptr() + HDR, cap, self.d_buf.data_ptr(), stream=self.stream.cuda_stream)
            else:
                self.ctx.match_device(d_img.data_ptr(), ROWS, COLS, COLS * 3, 3, THRESHOLD, self.d_buf.data_ptr() + HDR, cap,
                                      self.d_buf.data_ptr(), stream=self.stream.cuda_stream)

        def host_counts(self):
            """[world][B][2] = {n_matches, overflow} per rank and frame of the step"""
            return self.h_buf.numpy().reshape(world, BUF)[:, : 8 * B].copy().view(np.int32).reshape(world, B, 2)

        def host_records(self):
            """[world][B][cap] match records"""
            return self.h_buf.numpy().reshape(world, BUF)[:, HDR:].copy().view(MATCH_DTYPE).reshape(world, B, cap)

    slots = [Slot(None) for i in range(max(1, args.inflight))]
    ctx = slots[0].ctx
    stream = slots[0].stream
    torch.cuda.synchronize()
    step_no = [0]

    def step():
        slots[step_no[0] % len(slots)].run()
        step_no[0] += 1

    def fence():
        torch.cuda.synchronize()
        if collective:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if collective:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # one frame at a time on one stream (latency-bound figure), same K steps, outside the timed region
    single_ms = None
    if len(slots) > 1:
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            slots[0].run()
        fence()
        single_ms = (time.perf_counter() - t1) / args.steps * 1e3
    else:
        single_ms = elapsed / args.steps * 1e3

    # every slot must hold the same, stable match list (checked outside the timed region)
    ref_counts = None
    for rep in range(3):
        for sl in slots:
            sl.h_buf.zero_()
            sl.run()
            torch.cuda.synchronize()
            c = sl.host_counts()
            if ref_counts is None:
                ref_counts = c
            if not np.array_equal(c, ref_counts) or c[:, :, 0].min() <= 0:
                raise SystemExit(f"unstable match counts: {c.tolist()} vs {ref_counts.tolist()}")
    counts = slots[0].host_counts()
    if (counts[:, :, 1] != 0).any() or (counts[:, :, 0] > cap).any():
        raise SystemExit(f"match list overflow: {counts.tolist()}")
    recs = slots[0].host_records()
    if B > 1 and not collective:
        # every frame of the batch against the single-frame entry point on the same frame
        one_out = torch.zeros(cap * REC, dtype=torch.uint8, device=dev)
        one_cnt = torch.zeros(2, dtype=torch.int32, device=dev)
        ctx.set_result_mirror(0, 0)
        for b in range(B):
            ctx.match_device(d_img.data_ptr() + b * FRAME_BYTES, ROWS, COLS, COLS * 3, 3, THRESHOLD, one_out.data_ptr(), cap,
                             one_cnt.data_ptr(), stream=stream.cuda_stream)
            torch.cuda.synchronize()
            n1 = int(one_cnt.cpu().numpy()[0])
            single = capi.canonicalize(one_out.cpu().numpy().view(MATCH_DTYPE)[:n1].copy())
            batched = capi.canonicalize(recs[0, b, : counts[0, b, 0]].copy())
            if n1 != counts[0, b, 0] or single.tobytes() != batched.tobytes():
                raise SystemExit(f"frame {b} of the batch differs from its single-frame match list")
        ctx.set_result_mirror(slots[0].h_buf.data_ptr() + HDR, slots[0].h_buf.data_ptr())
    # frame 0 of the step: every rank's list, gathered
    matches = np.concatenate([recs[r, 0, : counts[r, 0, 0]] for r in range(world)])
    n_matches = len(capi.canonicalize(matches))

    # per-kernel durations: a second pass of the same steps on ONE slot (its stream runs the kernels back to back with
    # nothing synchronised in between, so a kernel's duration is its own: with two slots the kernels of the two steps
    # overlap on the GPU and stretch each other), timed with the dispatch packets' own start/stop timestamps
    # (hipExtLaunchKernelGGL events on the launch stream); kept out of the timed region above
    slots[0].ctx.set_profiling(True, accumulate=True)
    per_kernel = {}
    prof_steps = min(args.steps, 50)
    for rep in range(prof_steps):
        slots[0].run()
    fence()
    for name, ms in slots[0].ctx.timings():
        per_kernel.setdefault(name, []).append(ms)
    slots[0].ctx.set_profiling(False)
    n_cand, refine_bytes = ctx.stats()
    coarse_bytes = ctx.coarse_bytes()

    if rank == 0:
        # algorithmic bytes per launch (SURVEY.md 8d / DESIGN.md), by kernel
        npx = [(ROWS >> l) * (COLS >> l) for l in range(len(T_LEVELS))]
        alg = {
            # frame read + one-hot map written (+ the next level's image, written by the fused pyrDown);
            # a launch covers the B frames of the step (refinement bytes: frame 0's figure x B)
            "k_quantize": [B * (npx[0] * (3 + 1) + npx[1] * 3), B * npx[1] * (3 + 1)],
            "k_build_lm": [B * sum(n * 9 for n in npx)],
            "k_similarity_coarse": [B * coarse_bytes],
            "k_similarity_local": [B * refine_bytes],
        }
        kern = {}
        for name, v in per_kernel.items():
            launches = len(v) // prof_steps
            a = np.asarray(v).reshape(prof_steps, launches)
            kern[name] = {"ms_per_step": float(a.sum(axis=1).mean()), "launches": launches,
                          "avg_launch_us": float(a.mean() * 1e3),
                          # the launches of a step in order (k_quantize: pyramid level 0, level 1, ...)
                          "launch_us": [float(x) for x in a.mean(axis=0) * 1e3]}
            if name in alg:
                kern[name]["algorithmic_bytes_per_step"] = float(sum(alg[name]))
                kern[name]["achieved_GBps"] = float(sum(alg[name])) / (kern[name]["ms_per_step"] * 1e-3) / 1e9
        dom = max((k for k in kern if k in alg), key=lambda k: kern[k]["ms_per_step"])
        dom_bytes = float(sum(alg[dom])) / kern[dom]["launches"]
        dom_s = kern[dom]["avg_launch_us"] * 1e-6
        achieved = dom_bytes / dom_s / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get(dom)
            except Exception:
                traffic = None
        step_bytes = float(sum(sum(v) for v in alg.values()))
        total_templates = ts.n_templates
        value = total_templates * (ROWS * COLS / 1e6) * B * args.steps / elapsed
        out = {
            "metric": "templates*Mpixels/sec (whole Detector::match, frame resident in HBM)",
            "value": value,
            "unit": "templates*Mpixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": ("reference case1 test image (test/case1/test.png) centred on a black 1024x1024 BGR canvas; "
                     if args.frame == "case1" else
                     "reference case1 test image (test/case1/test.png) tiled over the whole 1024x1024 BGR canvas; ")
                    + "case1 rotation templates 0..359 (test/case1/test_templ.yaml)",
            "config": {
                "workload": "case1 on MI355X: 1024x1024x3 frames x 360 templates per GPU (131/71 features), "
                            f"pyramid T={{4,8}}, threshold 90, {B} frame(s) per step, every frame's match list gathered "
                            "to the host every step",
                "templates_total": total_templates,
                "templates_per_gpu": count,
                "frame": [ROWS, COLS, 3],
                "parallelism": f"template-shard x{world}" + (" + RCCL all-gather of match lists" if collective else ""),
                "frames_per_step": B,
                "us_per_frame": elapsed / args.steps / B * 1e6,
                "frames_in_flight": len(slots) * B,
                ("ms_per_step_one_frame_at_a_time" if B == 1 else "ms_per_step_one_batch_at_a_time"): single_ms,
                # SURVEY 8d's two times per frame, from the per-kernel pass (kernels alone on one stream):
                # t_match = all five kernels, t_templ = the template loop (coarse + refinement) only
                "t_match_kernels_us_per_frame": sum(v["ms_per_step"] for v in kern.values()) * 1e3 / B,
                "t_templ_kernels_us_per_frame": sum(v["ms_per_step"] for k, v in kern.items() if k.startswith("k_similarity")) * 1e3 / B,
                "matches_distinct": n_matches,
                "coarse_candidates_rank0": n_cand,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": dom,
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": dom_bytes,
                "avg_launch_us": kern[dom]["avg_launch_us"],
                "note": "per launch = the frames of one step; figures are the mean over this kernel's launches of a step "
                        "(k_quantize: one launch per pyramid level).  Per 1-Mpixel frame every kernel moves <= 10 MB from "
                        "HBM (<= 1.3 us at 8 TB/s); k_quantize is VALU/latency-bound, k_similarity_coarse streams its "
                        "72 MB per frame of algorithmic bytes from the L2-resident linear memories",
                "whole_step": {"algorithmic_bytes": step_bytes,
                               "achieved": step_bytes / (elapsed / args.steps) / 1e9,
                               "frac": step_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS},
            },
            "kernels": kern,
        }
        if world == 1 and not args.no_cpu_baseline:
            base_ts = ts.subset(range(first, first + count))
            out["cpu_baseline"] = cpu_baseline(base_ts, frame, args.cpu_budget)
        print(json.dumps(out))
    for sl in slots:
        sl.ctx.close()
    if collective:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
