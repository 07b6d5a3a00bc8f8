#!/usr/bin/env python3
"""bench.py — throughput of the LINE-2D match() hot path on MI355X.

A step = one pass of the hot path over one batch of frames that are already resident in HBM: for every frame one
whole Detector::match (line2Dup.cpp:1078-1150) — gradient quantisation -> pyramid -> spread/response/linearize ->
similarity over this rank's template shard -> 16x16 refinement -> match records — with the per-frame match lists
all-gathered over RCCL when N > 1 and stored in pinned host memory.  Metric: templates * Mpixels / s
(BASELINE.json), whole job: templates x Mpixels x frames per step x steps / time.

Default workload (BASELINE.json configs[1], "case1 on 1x MI355X"): 1024 x 1024 BGR frames -- SURVEY 8d's Stage-A scene
(random shapes + noise on every pixel) with the reference's case1 test image pasted at the centre, so that the figure
does not depend on constant regions of the canvas --, 360 case1 rotation templates (131 / 71 features) per GPU, pyramid
{4, 8}, threshold 90.
--batch frames per step (default 16: sbm_match_batch_device launches every kernel once for the batch; frame b is the
workload frame shifted 8*b columns) and --inflight independent slots (contexts + streams, default 4) used
round-robin; before the timed region a short probe picks the launch path (stream launches / hipGraph replay; all
slots / three / one: config.launch).  The same line also carries, as secondary, separately timed passes: the reference's own demo frame
(the test image on a black canvas, config.value_case1_canvas: 65 % constant pixels, the gradient kernel's best case), the
image tiled over the canvas (config.value_textured) and the scene without the object (config.value_stage_a).

Other BASELINE configurations (not the driver's line; --config):
  c3  2048^2, 3600 templates x 63/31 features          template loop on Stage-B maps, template-sharded
  c4  4096^2, 36 000 templates x 8191/4095 features    template loop on Stage-B maps, template-sharded
  c5  64 x (1920 x 1072) BGR frames, 1000 templates    whole match(), FRAME-sharded (rank r takes frames r::N)
--scaling weak (default: the per-GPU work is fixed) | strong (the total work is fixed and divided over the ranks).

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
VALU_ISSUE_CEILING = 0.6e12  # wave-instructions per second the chip issues of the integer mix these kernels are made of
                             # (v_dot2 / v_pk_* / v_perm / v_alignbit / DPP / v_bitop3: one per 4.1 cycles and SIMD; measured,
                             # tools/valu_rate.hip, profiles/r02_valu_issue_rates.txt)
THRESHOLD = 90.0
T_LEVELS = (4, 8)
PREFETCH = 256  # capacity (records) of the per-frame match list exchanged between ranks / sent to the host


# ---------------------------------------------------------------------------------------------------------------
# workloads
# ---------------------------------------------------------------------------------------------------------------
def case1_templates(n):
    from shape_based_matching_amd.templates import TemplateSet

    base = TemplateSet.load_npz(os.path.join(ROOT, "tests", "golden", "case1_templates.npz"))
    sets, k = [], 0
    while n > 0:
        take = min(n, 360)
        s = base.subset(range(take))
        s.class_ids = [f"test{k}"]
        sets.append(s)
        n -= take
        k += 1
    return TemplateSet.concat(sets)


def case1_frame(kind, rows, cols):
    """scene: SURVEY 8d's Stage-A scene (random filled shapes + noise in [-2, 2] on every pixel) with the reference's
    case1 test image pasted at the centre -- no constant region, and the trained object is there to be found;
    case1: the reference's test image centred on a black canvas (test.cpp:344-353 pads it the same way);
    tiled: the image repeated over the whole canvas (no constant region anywhere);
    stagea: SURVEY 8d's Stage-A input — black background, random filled rectangles / ellipses, noise in [-2, 2]."""
    from shape_based_matching_amd import synth

    img = np.load(os.path.join(ROOT, "tests", "golden", "case1_test_bgr.npz"))["bgr"]
    if kind == "tiled":
        reps = (-(-rows // img.shape[0]), -(-cols // img.shape[1]), 1)
        return np.ascontiguousarray(np.tile(img, reps)[:rows, :cols])
    if kind == "stagea":
        return synth.scene_bgr(1234, rows, cols)
    if kind == "scene":
        return synth.scene_with_object(1234, rows, cols, img)
    return synth.embed(img, rows, cols, (rows - img.shape[0]) // 2, (cols - img.shape[1]) // 2)


class Workload:
    """What a step does.  stage = "match": whole match() of `frames` (uint8 [B, rows, cols(, 3)]);
    stage = "templates": the template loop on Stage-B orientation maps that are uploaded once."""


def make_workload(args, world, rank=0):
    from shape_based_matching_amd import synth

    w = Workload()
    w.name, w.scaling = args.config, args.scaling
    w.shard = "templates"
    w.partition = "templates"
    w.maps = None
    if args.config == "case1":
        w.rows = w.cols = 1024
        w.ch, w.stage = 3, "match"
        per = args.templates or 360
        w.ts = case1_templates(per * world if args.scaling == "weak" else per)
        w.batch = max(1, args.batch)
        part = args.partition
        if part == "auto":
            part = "frames" if (args.scaling == "strong" and world > 1 and w.batch % world == 0) else "templates"
        w.partition = part
        if part == "frames":
            w.shard = "frames"
            w.total_frames = w.batch * world if args.scaling == "weak" else w.batch
        frame = case1_frame(args.frame, w.rows, w.cols)
        w.frames = np.stack([np.roll(frame, 8 * b, axis=1) for b in range(w.total_frames if part == "frames" else w.batch)])
        w.desc = (f"case1 on MI355X: 1024x1024x3 frames x {per} templates {'per GPU' if args.scaling == 'weak' else 'in total'} "
                  f"(131/71 features), pyramid T={{4,8}}, threshold 90, {w.batch} frame(s) per step, every frame's match list "
                  "gathered to the host every step")
        w.data = {"scene": "SURVEY 8d Stage-A scene (random shapes + noise on every pixel) 1024x1024 BGR with the reference case1 "
                           "test image (test/case1/test.png) pasted at the centre; ",
                  "case1": "reference case1 test image (test/case1/test.png) centred on a black 1024x1024 BGR canvas; ",
                  "tiled": "reference case1 test image (test/case1/test.png) tiled over the whole 1024x1024 BGR canvas; ",
                  "stagea": "SURVEY 8d Stage-A scene (random shapes + noise) 1024x1024 BGR; "}[args.frame] + \
            "case1 rotation templates 0..359 (test/case1/test_templ.yaml)"
    elif args.config in ("c3", "c4"):
        w.rows = w.cols = 2048 if args.config == "c3" else 4096
        w.ch, w.stage, w.batch = 1, "templates", 1
        nf, box, total = ([63, 31], 260, 3600) if args.config == "c3" else ([8191, 4095], 1024, 36000)
        total = args.templates or total
        n = total * world if args.scaling == "weak" else total
        # every template has the same number of features and positions: equal contiguous ranges are work-balanced, and a
        # rank generates only its own shard (stage_b_fixed: per-template random streams, a fixed number of plants)
        w.n_total = n
        w.range = (rank * n // world, (rank + 1) * n // world - rank * n // world)
        plants = 32 if args.config == "c3" else 16
        w.maps, w.ts = synth.stage_b_fixed(1234, w.rows, w.cols, T_LEVELS, n, nf, templ_size=box, n_plants=plants,
                                           first=w.range[0], count=w.range[1])
        w.frames = None
        w.desc = (f"BASELINE config {args.config[1]}: {w.rows}x{w.cols} Stage-B orientation maps (2 % one-hot density + {plants} "
                  f"planted templates), {n} templates x {nf[0]}/{nf[1]} features, template loop (coarse pass + refinement) per step")
        w.data = "synthetic Stage-B inputs (SURVEY 8d generator with a fixed number of plants: synth.stage_b_fixed), seed 1234"
    else:  # c5
        w.rows, w.cols, w.ch, w.stage = 1072, 1920, 3, "match"
        w.shard = "frames"
        total_frames = args.batch if args.batch != 16 else 64
        w.total_frames = total_frames * world if args.scaling == "weak" else total_frames
        w.ts = case1_templates(args.templates or 1000)
        frame = case1_frame("tiled", w.rows, w.cols)
        w.frames = np.stack([np.roll(frame, 8 * b, axis=1) for b in range(w.total_frames)])
        w.batch = None  # per rank, set by the caller
        w.desc = (f"BASELINE config 5: {w.total_frames} x (1920x1072x3) frames (the 1080p frame cropped to multiples of 16), "
                  f"{w.ts.n_templates} templates (131/71 features), frame-sharded: rank r matches frames r::N, one gather of the lists")
        w.data = "reference case1 test image tiled over 1920x1072; case1 rotation templates repeated to 1000"
    return w


def usable_cores():
    """CPUs this process may actually use: affinity mask, capped by the cgroup CPU quota (a GPU box shows all 256 logical
    cores of the host but grants a share of them; threads beyond the share only thrash)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    try:
        import psutil

        phys = psutil.cpu_count(logical=False)
        if phys:
            n = min(n, phys)
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(ts, frame, budget_s: float = 12.0):
    """The CPU oracle (a port of the reference algorithm) timed on this host on the same frame and templates, in the
    reference's shape: the pyramid build over row bands (what OpenCV's parallel_for_ does inside the calls of
    line2Dup.cpp:1084-1120) and the OpenMP loop over templates (line2Dup.cpp:1166-1170, static schedule), both with the
    same thread count.  The count is chosen by measurement among powers of two up to the cores this process may use.
    Returns (json dict, match list)."""
    from oracle import oracle as O

    ncpu = os.cpu_count() or 1
    usable = usable_cores()
    last = [None]

    def run(threads, reps):
        O.set_build_threads(threads)
        t0 = time.perf_counter()
        for _ in range(reps):
            pyr = O.Pyramid.build(frame, list(T_LEVELS), 30.0)
            last[0] = pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, THRESHOLD, n_threads=threads)
            pyr.free()
        O.set_build_threads(1)
        return (time.perf_counter() - t0) / reps

    cands = sorted({1, usable} | {k for k in (2, 4, 8, 16, 32, 64) if k <= usable})
    run(1, 1)  # page in
    probe = {k: run(k, 2) for k in cands}
    threads = min(probe, key=probe.get)
    reps = max(3, int(budget_s / probe[threads]))
    per = run(threads, reps)
    value = ts.n_templates * (frame.shape[0] * frame.shape[1] / 1e6) / per
    return {
        "value": value,
        "unit": "templates*Mpixels/s",
        "cores": threads,
        "kind": "port",
        "sample": f"oracle/sbm_oracle.c (plain C, gcc -O3 -mavx2 -fopenmp -ffp-contract=off: compiler-vectorised, no hand-written SIMD); "
                  f"{reps} full match() calls of the bench frame with {ts.n_templates} templates "
                  f"({per * 1e3:.1f} ms each, {len(last[0])} raw matches) on {threads} thread(s): row-band parallel pyramid "
                  f"build + OpenMP template loop; host has {ncpu} logical cores, {usable} usable by this process",
        "ms_per_match": per * 1e3,
        "ms_per_match_1_thread": probe[1] * 1e3,
        "ms_per_match_by_threads": {str(k): round(v * 1e3, 2) for k, v in probe.items()},
    }, last[0]


def spawn_ranks(n_gpus):
    """`python3 bench.py --gpus N` without a launcher: start the N ranks ourselves, one process per GPU, BEFORE anything in
    this process imports torch or touches a GPU (a process that has initialised the GPU must not be replaced or forked).
    Same environment a `torch.distributed.run --nnodes=1 --nproc-per-node N` child sees (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_*); rank 0's stdout -- the one JSON line -- is forwarded verbatim, every rank's stderr goes to ours.  Returns the
    exit code: 0 only if every rank exited with 0."""
    import socket
    import subprocess

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n_gpus):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n_gpus), "LOCAL_WORLD_SIZE": str(n_gpus),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "SBM_BENCH_SPAWNED": "1"})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    print(f"[bench] spawned {n_gpus} ranks (pids {[p.pid for p in procs]}), rendezvous 127.0.0.1:{port}", file=sys.stderr, flush=True)
    import threading

    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)  # until rank 0 closes its stdout
    reader.start()
    # A rank that dies early (no GPU, a failed communicator) leaves the others waiting in a rendezvous or a collective for
    # minutes: once any rank has exited with an error the rest get a short grace period and are then killed -- exactly the
    # processes started here, by pid
    rcs = [None] * n_gpus
    failed_at = None
    while any(rc is None for rc in rcs):
        for r, p in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = p.poll()
                if rcs[r] not in (None, 0) and failed_at is None:
                    failed_at = time.time()
        if failed_at is not None and time.time() - failed_at > 20.0:
            for r, p in enumerate(procs):
                if rcs[r] is None:
                    p.kill()
                    rcs[r] = p.wait()
                    print(f"[bench] rank {r} (pid {p.pid}) was still running 20 s after another rank failed: killed", file=sys.stderr)
        time.sleep(0.05)
    reader.join(timeout=5.0)
    sys.stdout.buffer.write(out0[0] if out0 else b"")
    sys.stdout.flush()
    print(f"[bench] rank exit codes: {rcs}", file=sys.stderr, flush=True)
    # the code of a rank that failed by itself, if any; 1 if ranks only died by signal; 0 if all succeeded
    return next((rc for rc in rcs if 0 < rc < 256), 1 if any(rc != 0 for rc in rcs) else 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default: 1000 (case1), 200 (c3), 50 (c5), 10 (c4)")
    ap.add_argument("--warmup", type=int, default=None, help="default: a tenth of the steps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    ap.add_argument("--force-collective", action="store_true",
                    help="rehearsal: run the N>1 code path (RCCL all-gathers + host copy) with whatever world size")
    ap.add_argument("--inflight", type=int, default=0,
                    help="batches in flight per GPU: independent engine contexts + HIP streams used round-robin "
                         "(1 = strictly one batch at a time; the single-stream figure is always reported too).  Default: 4 "
                         "for case1, 1 for c4, 2 otherwise")
    ap.add_argument("--input-ring", type=int, default=8,
                    help="distinct device copies of the step's frames used round-robin (1 = every step reads the same buffer, "
                         "which then lives in the Infinity Cache)")
    ap.add_argument("--batch", type=int, default=16,
                    help="frames per step: a step is one sbm_match_batch_device call over this many frames (distinct "
                         "horizontal shifts of the workload frame); 1 = one sbm_match_device call per step")
    ap.add_argument("--frame", choices=("scene", "case1", "tiled", "stagea"), default="scene",
                    help="scene (default): SURVEY 8d's shapes + noise scene with the case1 test image pasted in (no constant "
                         "region: a content-independent figure); case1: the reference's test image centred on a black canvas "
                         "(65 %% constant: the gradient kernel's constant-row shortcut applies); tiled: the image repeated over "
                         "the whole canvas; stagea: the scene without the object")
    ap.add_argument("--config", choices=("case1", "c3", "c4", "c5"), default="case1")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--templates", type=int, default=0, help="override the configuration's template count")
    ap.add_argument("--partition", choices=("auto", "templates", "frames", "bands"), default="auto",
                    help="how a step is divided over the ranks (case1): templates = contiguous template ranges, every rank "
                         "builds the whole pyramid (the reference's OpenMP loop, line2Dup.cpp:1166-1170); frames = the frames of "
                         "the batch dealt over the ranks, all templates each (no exchange but the match lists); bands = "
                         "build-sharded: row band r of every level's orientation map on rank r + all-gather of the bands + "
                         "template ranges (sbm_match_batch_device_banded).  auto: templates for weak scaling, frames for "
                         "strong scaling of a batch that divides over the ranks")
    ap.add_argument("--bands", type=int, default=0,
                    help="one GPU only: run every step through the band-sharded entry point with this many row bands (all "
                         "computed here, one launch per band and level) -- the rehearsal of --partition bands")
    ap.add_argument("--no-strong-estimate", action="store_true", help="skip the measured strong-scaling estimate (N = 1)")
    ap.add_argument("--no-extra-frames", action="store_true", help="skip the secondary (textured / Stage-A) passes")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))  # no launcher around us: this process only starts the ranks and forwards rank 0's line
    if os.environ.get("SBM_BENCH_SELFTEST") == "rank1_fails_rank0_waits" and os.environ.get("SBM_BENCH_SPAWNED"):
        # test hook of the launcher's supervision (tests/test_bench_spawn.py): no GPU work at all
        if os.environ.get("RANK") == "1":
            raise SystemExit(3)
        time.sleep(300)
    if args.steps is None:
        args.steps = {"case1": 1000, "c3": 200, "c4": 10, "c5": 50}[args.config]
    if args.warmup is None:
        args.warmup = max(2, args.steps // 10)
    if args.inflight <= 0:
        # c4: one coarse launch (26 ms for 4 500 templates) fills the GPU for its whole length; a second one in flight only
        # contends with it (32.5 ms per step with two slots against 26.2 with one)
        args.inflight = {"case1": 4, "c4": 1}.get(args.config, 2)
    # HIP spreads a process's streams over GPU_MAX_HW_QUEUES hardware queues (default 4) and two streams that share a
    # queue run their kernels in order.  Rounds 1 and 2 saw four slots + the null stream land two slots on one queue in
    # some sessions (14.5 us per frame against 7.4 with 8 queues) and used three.  Round 3: with the
    # waiting kernels' waves at a raised issue priority four slots measure 109.5 - 109.8 us per step against 112.6 -
    # 113.1 with three, in every one of six processes (tools/r03_after_prio.sh); the probe below still tries three.

    # stdout carries exactly ONE line, the JSON record: libraries that write to file descriptor 1 themselves (RCCL prints
    # a version banner there when a communicator is created) are sent to stderr for the whole run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    from shape_based_matching_amd import capi, sharding
    from shape_based_matching_amd.templates import MATCH_DTYPE

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher's WORLD_SIZE is {world}: make them agree (or unset WORLD_SIZE and "
                         "let bench.py start its own ranks)")
    if not torch.cuda.is_available():
        raise SystemExit(f"bench.py rank {rank} of {world}: needs an MI355X: no GPU visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    collective = world > 1 or args.force_collective
    if collective:
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"] = "127.0.0.1"
            os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    wl = make_workload(args, world, rank)
    ts = wl.ts
    ROWS, COLS, CH = wl.rows, wl.cols, wl.ch
    if wl.maps is not None:  # Stage-B template loops: this rank generated exactly its shard
        first, count = 0, ts.n_templates
        my_frames = None
    elif wl.shard == "templates":
        first, count = sharding.partition(sharding.coarse_work(ts, ROWS, COLS, T_LEVELS), world)[rank]
        my_frames = np.arange(wl.batch) if wl.frames is not None else None
    else:  # frames dealt rank::world, templates replicated
        first, count = 0, ts.n_templates
        my_frames = sharding.frame_shard(wl.total_frames, world, rank)
    B = len(my_frames) if my_frames is not None else 1
    cap = (PREFETCH if args.config == "case1" else 2048) if wl.stage == "match" else 4096
    FRAME_BYTES = ROWS * COLS * CH
    d_img = torch.from_numpy(np.ascontiguousarray(wl.frames[my_frames])).to(dev) if wl.frames is not None else None

    REC = MATCH_DTYPE.itemsize
    # per rank: B {n_matches, overflow} int32 pairs (padded to 16 bytes) in front of B blocks of cap records:
    # one buffer, one collective, one copy
    HDR = (8 * B + 15) // 16 * 16
    BUF = HDR + B * cap * REC
    # ONE exchange path whatever the configuration: the library's own RCCL communicator (sbm_comm_init), its ncclAllGather
    # issued by the sharded entry point on the kernels' stream.  torch.distributed (backend nccl = RCCL as well) only
    # carries the 128-byte communicator id to the ranks and the barrier / max-over-ranks around the timed region.
    n_bands = world if (world > 1 and wl.partition == "bands") else (args.bands if world == 1 else 0)
    banded = wl.stage == "match" and n_bands > 0 and args.config == "case1"
    if banded and B < 1:
        raise SystemExit("--partition bands needs a batch")

    img_sel = [None]
    ring_state = [None, 0]  # [the ring of input buffers (set below), calls so far]

    class Slot:
        """one step in flight: its own engine context (device buffers), stream and result buffers"""

        def __init__(self):
            self.ctx = capi.Context(T=T_LEVELS, weak_threshold=30.0, device_id=local_rank,
                                    max_candidates=0 if wl.stage == "match" else 1 << 22)
            self.ctx.upload_templates(ts)
            self.ctx.select_range(first, count)
            if os.environ.get("SBM_GRAPH", "0") not in ("", "0"):
                self.ctx.set_graph_mode(True)
            if wl.maps is not None:
                for l in range(len(T_LEVELS)):
                    self.ctx.set_quantized(l, wl.maps[l])
            # always an explicit stream: handle 0 would mean "the context's own stream" to the C ABI and
            # the result copies below must be ordered after the kernels
            self.stream = torch.cuda.Stream(device=dev)
            self.d_buf = torch.zeros(BUF, dtype=torch.uint8, device=dev)          # this rank: header + records
            self.g_buf = torch.zeros(world * BUF, dtype=torch.uint8, device=dev)  # all ranks, gathered
            self.h_buf = torch.zeros(world * BUF, dtype=torch.uint8).pin_memory()
            if not collective and not banded:
                # single GPU: the last kernel stores the match list straight into pinned host memory
                self.ctx.set_result_mirror(self.h_buf.data_ptr() + HDR, self.h_buf.data_ptr())
            elif collective:
                # one RCCL communicator per slot, its id made by rank 0 and broadcast.  Every rank takes part in both
                # broadcasts whatever happened to it before; whether ALL ranks got a communicator is agreed on below.
                self.comm_error = None
                uid = torch.zeros(128, dtype=torch.uint8, device=dev)
                if rank == 0:
                    try:
                        uid.copy_(torch.frombuffer(bytearray(capi.Context.comm_unique_id()), dtype=torch.uint8))
                    except (capi.SbmError, OSError) as e:
                        self.comm_error = f"no RCCL unique id from the library ({e})"
                ok = torch.tensor([0 if self.comm_error else 1], dtype=torch.int32, device=dev)
                dist.broadcast(ok, src=0)
                dist.broadcast(uid, src=0)
                if not int(ok.item()):
                    self.comm_error = self.comm_error or "rank 0 could not make a communicator id"
                else:
                    try:
                        self.ctx.comm_init(world, rank, bytes(uid.cpu().numpy().tobytes()))
                    except (capi.SbmError, OSError) as e:
                        self.comm_error = f"sbm_comm_init failed ({e})"

        def run(self):
            s = self.stream.cuda_stream
            if ring_state[0] is not None:  # every call reads the next buffer of the input ring
                img_sel[0] = ring_state[0][ring_state[1] % len(ring_state[0])]
                ring_state[1] += 1
            if wl.stage == "templates" and collective:
                # template-loop configurations: this rank's template range on the resident pyramid + the exchange step
                self.ctx.match_templates_device_sharded(THRESHOLD, self.d_buf.data_ptr(), cap, self.g_buf.data_ptr(),
                                                        gathered_mirror=self.h_buf.data_ptr(), stream=s)
            elif wl.stage == "templates":
                self.ctx.match_templates_device(THRESHOLD, self.d_buf.data_ptr() + HDR, cap, self.d_buf.data_ptr(), stream=s)
            elif banded:
                # build-sharded step: row bands of the gradient stage + all-gather of the orientation maps + template
                # ranges + gather of the lists (one GPU: all bands here, one launch per band and level)
                self.ctx.match_batch_device_banded(img_sel[0].data_ptr(), FRAME_BYTES, B, ROWS, COLS, COLS * CH, CH, THRESHOLD,
                                                   self.d_buf.data_ptr(), cap, self.g_buf.data_ptr() if collective else 0,
                                                   gathered_mirror=self.h_buf.data_ptr(), n_bands=0 if world > 1 else n_bands,
                                                   stream=s)
            elif collective and B > 1:
                self.ctx.match_batch_device_sharded(img_sel[0].data_ptr(), FRAME_BYTES, B, ROWS, COLS, COLS * CH, CH, THRESHOLD,
                                                    self.d_buf.data_ptr(), cap, self.g_buf.data_ptr(),
                                                    gathered_mirror=self.h_buf.data_ptr(), stream=s)
            elif collective:
                # match of this rank's shard + the exchange step (ncclAllGather over xGMI, issued by the library on the
                # same stream) + copy of the gathered lists into pinned host memory
                self.ctx.match_device_sharded(img_sel[0].data_ptr(), ROWS, COLS, COLS * CH, CH, THRESHOLD, self.d_buf.data_ptr(), cap,
                                              self.g_buf.data_ptr(), gathered_mirror=self.h_buf.data_ptr(), stream=s)
            elif B > 1:
                self.ctx.match_batch_device(img_sel[0].data_ptr(), FRAME_BYTES, B, ROWS, COLS, COLS * CH, CH, THRESHOLD,
                                            self.d_buf.data_ptr() + HDR, cap, self.d_buf.data_ptr(), stream=s)
            else:
                self.ctx.match_device(img_sel[0].data_ptr(), ROWS, COLS, COLS * CH, CH, THRESHOLD, self.d_buf.data_ptr() + HDR, cap,
                                      self.d_buf.data_ptr(), stream=s)

        def host_counts(self):
            """[world][B][2] = {n_matches, overflow} per rank and frame of the step"""
            return self.h_buf.numpy().reshape(world, BUF)[:, : 8 * B].copy().view(np.int32).reshape(world, B, 2)

        def host_records(self):
            """[world][B][cap] match records"""
            return self.h_buf.numpy().reshape(world, BUF)[:, HDR:].copy().view(MATCH_DTYPE).reshape(world, B, cap)

    exchange_path = ["library (ncclAllGather on the kernels' stream)" if collective else "none"]
    slots = [Slot() for _ in range(max(1, args.inflight))]
    ctx = slots[0].ctx
    n_ranks_seen = 1
    if collective:
        # every rank must hold a communicator on every slot: agree on it, and stop all ranks together if one does not
        errs = [sl.comm_error for sl in slots if sl.comm_error]
        ok = torch.tensor([0 if errs else 1], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            raise SystemExit(f"bench.py rank {rank} of {world}: the library's RCCL communicator could not be set up on every rank"
                             + (f" (here: {errs[0]})" if errs else ""))
        n_ranks_seen = ctx.comm_count()  # as RCCL reports it
    if not os.environ.get("SBM_BENCH_LATENCY_SIZING"):
        for sl in slots:  # the slots' batches are in flight together: size the launches for throughput (and, with >= 2, let
            sl.ctx.set_pipeline_depth(len(slots))  # the library replay captured graphs: its default, sbm_set_graph_mode)
    torch.cuda.synchronize()
    step_no = [0]
    # The steps read their frames from a RING of distinct device buffers (same contents): a stream of frames arrives from
    # HBM, and with one buffer re-used by every step the 50 MB of a 16-frame batch stay in the 256 MB Infinity Cache
    # (measured: 109.1 us per step with one buffer, 111.1 - 112.0 with 4, 8 or 12 -- tools/r03_ring.sh).  8 x 50 MB by
    # default: more than that cache holds beside the engine's own buffers.
    ring = [d_img] + [d_img.clone() for _ in range(max(1, args.input_ring) - 1)] if d_img is not None else [None]
    img_sel[0] = ring[0]
    ring_state[0] = ring if d_img is not None else None

    def step():
        slots[step_no[0] % len(slots)].run()
        step_no[0] += 1

    def fence():
        torch.cuda.synchronize()
        if collective:
            dist.barrier()
            torch.cuda.synchronize()

    last_issue_s = [0.0]

    def timed(n_warm, n_steps, which=None):
        """K steps bracketed by barrier + synchronize; max over ranks"""
        run = step if which is None else which
        for _ in range(n_warm):
            run()
        fence()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            run()
        last_issue_s[0] = time.perf_counter() - t0  # host time to ENQUEUE the steps (nothing waited for yet)
        fence()
        el = time.perf_counter() - t0
        if collective:
            tt = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el

    # Launch configuration, chosen by measurement before the timed region.  The default -- stream launches, all slots
    # round-robin -- is the fastest when nothing is wrong (6.98 us per frame against 7.23 with hipGraph replay and 10.4
    # with one batch at a time).  One session in about forty on this pool ran the multi-slot stream path 2.6x slower
    # with unchanged kernel times (launch / dispatch latency outside the engine); a 40-step probe of the alternatives
    # the engine offers anyway (replay of the captured hipGraph: one host call per step; a single slot) costs a few
    # milliseconds and keeps such a session from deciding the figure.
    active = [slots]

    def step():  # noqa: F811 -- the round-robin over the slots in use
        a = active[0]
        a[step_no[0] % len(a)].run()
        step_no[0] += 1

    def probe(n=40):
        # warm-up: every (slot, input buffer) pair once -- in graph mode each pair is a capture of its own, which must not
        # fall into the timed steps
        return timed(max(2, len(ring)) * len(active[0]), n) / n * 1e6

    launch = {"path": "stream launches", "slots": len(slots)}
    graph_env = os.environ.get("SBM_GRAPH", "0") not in ("", "0")
    if graph_env:
        launch["path"] = "hipGraph replay (SBM_GRAPH)"
    elif wl.stage != "match" or B == 1 or banded:
        # nothing to choose: the library's own default (auto: captured-graph replay of the template loop / the batch once several
        # calls are in flight and an argument tuple repeats; plain launches for one frame at a time)
        launch["path"] = "library default (sbm_set_graph_mode auto)"
        if len(slots) > 1 and not os.environ.get("SBM_BENCH_NO_ADAPT"):
            # ... but how many calls to keep in flight is the caller's choice: some processes on this pool run launches that
            # alternate over several streams slowly, and a 20 us template loop then steps faster from
            # one slot (c3: 42 us per step with two slots against 27 with one in such a process, 20.5 against 26 otherwise)
            t_all = probe()
            active[0] = slots[:1]
            t_one = probe()
            launch["probe_us_per_step"] = {f"{len(slots)} slots": round(t_all, 1), "one slot": round(t_one, 1)}
            if t_one < 0.95 * t_all:
                launch["slots"] = 1
            active[0] = slots[: launch["slots"]]
    elif not os.environ.get("SBM_BENCH_NO_ADAPT"):
        for sl in slots:
            sl.ctx.set_graph_mode(False)  # the library's default with several batches in flight is replay: opt out for the A/B
        t_stream = probe()
        issue_stream = last_issue_s[0] / 40 * 1e6
        # a context keeps 16 captured graphs (one per argument tuple = input buffer here): a longer ring of input buffers
        # would evict and re-capture on every step, so replay is not a candidate then
        graph_ok = len(ring) <= 16
        for sl in slots:
            sl.ctx.set_graph_mode(graph_ok)
        t_graph = probe() if graph_ok else float("inf")
        launch["probe_us_per_step"] = {"stream launches": round(t_stream, 1), "hipGraph replay": round(t_graph, 1) if graph_ok else None}
        # host time spent inside the launch calls of a step (no synchronisation): tells a slow GPU from a blocking host
        launch["probe_host_enqueue_us_per_step"] = {"stream launches": round(issue_stream, 1),
                                                    "hipGraph replay": round(last_issue_s[0] / 40 * 1e6, 1)}
        # Replay of the captured graphs unless stream launches are clearly faster: with four slots the two measure the same
        # (109.3 - 110.4 against 109.7 - 111.1 us per step, six processes each: tools/r03_graph_vs_stream.sh), and about one
        # process in eight runs its stream launches in a slow multi-stream mode (0.5 - 1.8 ms per step; mildly, 5 %, in
        # others) that replay does not have
        prefer_stream = os.environ.get("SBM_BENCH_PREFER") == "stream"  # A/B of the tie rule
        if t_stream < 0.95 * t_graph or (prefer_stream and t_graph >= 0.95 * t_stream):
            for sl in slots:
                sl.ctx.set_graph_mode(False)
        else:
            launch["path"] = "hipGraph replay"
        # fewer slots: three (four slot streams and the null stream are five streams on the stock runtime's four hardware
        # queues; a session in which two slots land on one queue is slower with four than with three), one
        best = min(t_stream, t_graph)
        for n_try, label in ((3, "three slots"), (1, "one slot")):
            if len(slots) <= n_try:
                continue
            active[0] = slots[:n_try]
            t_n = probe()
            launch["probe_us_per_step"][label] = round(t_n, 1)
            if t_n < 0.95 * best:
                launch["slots"] = n_try
                best = t_n
        active[0] = slots[: launch["slots"]]

    elapsed = timed(args.warmup, args.steps)

    # one step at a time on one stream (latency-bound figure), same K steps, outside the timed region
    if len(slots) > 1:
        slots[0].ctx.set_pipeline_depth(1)  # one batch in flight: latency sizing
        single_ms = timed(2, args.steps, slots[0].run) / args.steps * 1e3
        if wl.stage == "match" and not os.environ.get("SBM_BENCH_LATENCY_SIZING"):
            slots[0].ctx.set_pipeline_depth(len(slots))
    else:
        single_ms = elapsed / args.steps * 1e3

    # every slot must hold the same, stable match list (checked outside the timed region)
    ref_counts = None
    for rep in range(3):
        for sl in slots:
            sl.h_buf.zero_()
            sl.run()
            fence()
            c = sl.host_counts()
            if ref_counts is None:
                ref_counts = c
            if not np.array_equal(c, ref_counts):
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
        for b in range(min(B, 16)):
            ctx.match_device(d_img.data_ptr() + b * FRAME_BYTES, ROWS, COLS, COLS * CH, CH, THRESHOLD, one_out.data_ptr(), cap,
                             one_cnt.data_ptr(), stream=slots[0].stream.cuda_stream)
            torch.cuda.synchronize()
            n1 = int(one_cnt.cpu().numpy()[0])
            single = capi.canonicalize(one_out.cpu().numpy().view(MATCH_DTYPE)[:n1].copy())
            batched = capi.canonicalize(recs[0, b, : counts[0, b, 0]].copy())
            if n1 != counts[0, b, 0] or single.tobytes() != batched.tobytes():
                raise SystemExit(f"frame {b} of the batch differs from its single-frame match list")
        if not banded:
            ctx.set_result_mirror(slots[0].h_buf.data_ptr() + HDR, slots[0].h_buf.data_ptr())
    # frame 0 of the step: every rank's list, gathered (frame-sharded: rank 0's first frame)
    src_ranks = range(world) if wl.shard == "templates" else range(1)
    matches = np.concatenate([recs[r, 0, : counts[r, 0, 0]] for r in src_ranks])
    n_matches = len(capi.canonicalize(matches.copy()))

    # per-kernel durations: a second pass of the same steps on ONE slot (its stream runs the kernels back to back with
    # nothing synchronised in between, so a kernel's duration is its own: with two slots the kernels of the two steps
    # overlap on the GPU and stretch each other), timed with the dispatch packets' own start/stop timestamps
    # (hipExtLaunchKernelGGL events on the launch stream); kept out of the timed region above
    def kernel_pass(n):
        # one batch at a time on one slot: size the launches for that (the hint is the number of batches in flight)
        slots[0].ctx.set_pipeline_depth(1)
        slots[0].run()
        fence()
        slots[0].ctx.set_profiling(True, accumulate=True)
        per = {}
        for _ in range(n):
            slots[0].run()
        fence()
        for name, ms in slots[0].ctx.timings():
            per.setdefault(name, []).append(ms)
        slots[0].ctx.set_profiling(False)
        if wl.stage == "match" and not os.environ.get("SBM_BENCH_LATENCY_SIZING"):
            slots[0].ctx.set_pipeline_depth(len(slots))
        out = {}
        for name, v in per.items():
            launches = len(v) // n
            a = np.asarray(v).reshape(n, launches)
            out[name] = {"ms_per_step": float(a.sum(axis=1).mean()), "launches": launches, "avg_launch_us": float(a.mean() * 1e3),
                         # the launches of a step in order (k_quantize: pyramid level 0, level 1, ...)
                         "launch_us": [float(x) for x in a.mean(axis=0) * 1e3]}
        return out

    prof_steps = min(args.steps, 50)
    kern = kernel_pass(prof_steps)
    n_cand, refine_bytes = ctx.stats()
    coarse_bytes = ctx.coarse_bytes()

    # Strong scaling of THIS step (the same B frames x the same templates divided over N GPUs), estimated from what one
    # rank of an N-GPU job would run, measured here on one GPU.  Three partitions:
    #   templates  contiguous template ranges, the whole pyramid built on every rank (the reference's OpenMP loop)
    #   bands      build-sharded: the rank's row band of every orientation map (all N band launches run here, the rank's
    #              time is the slowest band's), whole linear memories, 1 / N of the templates; + a MODEL of the all-gather
    #              of the maps (it cannot be measured on one GPU)
    #   frames     B / N frames, all templates: nothing exchanged but the match lists
    # Every figure is the one-batch-at-a-time kernel time of a rank's share (per-kernel timestamps), so it compares with
    # config.ms_per_step_one_batch_at_a_time; "speedup" = that / the rank's time (+ the modelled exchange for bands).
    strong = None
    if (world == 1 and args.config == "case1" and wl.stage == "match" and B >= 8 and not banded and not args.no_strong_estimate
            and not args.no_extra_frames):
        XGMI_LINK_GBS = 50.0   # assumed payload rate of ONE xGMI link in one direction (peak 76.5 of the 153 GB/s pair)
        XGMI_LATENCY_US = 12.0  # assumed fixed cost of one grouped all-gather launch over 8 ranks
        sl = slots[0]

        def kernel_us(run, n=20):
            sl.ctx.set_profiling(True, accumulate=True)
            for _ in range(n):
                run()
            fence()
            per = {}
            for name, ms in sl.ctx.timings():
                per.setdefault(name, []).append(ms * 1e3)
            sl.ctx.set_profiling(False)
            return {k: np.asarray(v).reshape(n, -1).mean(axis=0) for k, v in per.items()}

        sl.ctx.set_pipeline_depth(1)  # every figure below is one batch at a time
        base = kernel_us(sl.run)
        t1 = float(sum(v.sum() for v in base.values()))
        strong = {"one_gpu_kernels_us_per_step": t1, "assumed_xgmi_link_GBps": XGMI_LINK_GBS, "assumed_all_gather_latency_us": XGMI_LATENCY_US,
                  "note": "kernel time of one rank's share of the same step, measured on this GPU (one batch at a time); the "
                          "bands row adds a modelled all-gather of the orientation maps (1 byte per pixel and level, every "
                          "peer's band over its own xGMI link)"}
        s0 = sl.stream.cuda_stream
        out_p, cnt_p = sl.d_buf.data_ptr() + HDR, sl.d_buf.data_ptr()
        sl.ctx.set_result_mirror(0, 0)
        for n in (2, 4, 8):
            if B % n:
                continue
            e = {}
            # frames: B / n frames, all templates
            kf = kernel_us(lambda: sl.ctx.match_batch_device(d_img.data_ptr(), FRAME_BYTES, B // n, ROWS, COLS, COLS * CH, CH, THRESHOLD,
                                                             out_p, cap, cnt_p, stream=s0))
            tf = float(sum(v.sum() for v in kf.values()))
            e["frames"] = {"frames_per_rank": B // n, "rank_kernels_us": tf, "speedup": t1 / tf}
            # templates: the rank's template range, whole build
            parts = sharding.partition(sharding.coarse_work(ts, ROWS, COLS, T_LEVELS), n)
            sl.ctx.select_range(*parts[0])
            kt = kernel_us(sl.run)
            tt = float(sum(v.sum() for v in kt.values()))
            e["templates"] = {"templates_per_rank": parts[0][1], "rank_kernels_us": tt, "speedup": t1 / tt}
            # bands: all n band launches per level run here; a rank's gradient time is its slowest band's
            kb = kernel_us(lambda: sl.ctx.match_batch_device_banded(d_img.data_ptr(), FRAME_BYTES, B, ROWS, COLS, COLS * CH, CH, THRESHOLD,
                                                                    sl.d_buf.data_ptr(), cap, n_bands=n, stream=s0))
            q = kb["k_quantize"].reshape(len(T_LEVELS), n)
            tq = float(q.max(axis=1).sum())
            rest = float(sum(v.sum() for k, v in kb.items() if k != "k_quantize"))
            band_bytes = B * sum((ROWS >> l) * (COLS >> l) for l in range(len(T_LEVELS))) / n
            tx = XGMI_LATENCY_US + band_bytes / (XGMI_LINK_GBS * 1e3)
            e["bands"] = {"gradient_us_slowest_band": tq, "gradient_us_per_band": [[round(float(x), 2) for x in r] for r in q],
                          "other_kernels_us": rest, "all_gather_model_us": tx, "map_bytes_received_per_rank": band_bytes * (n - 1),
                          "rank_us_exchange_exposed": tq + rest + tx, "speedup_exchange_exposed": t1 / (tq + rest + tx),
                          "rank_us_exchange_hidden": max(tq + rest, tx), "speedup_exchange_hidden": t1 / max(tq + rest, tx)}
            sl.ctx.select_range(first, count)
            strong[str(n)] = e
        sl.ctx.set_result_mirror(sl.h_buf.data_ptr() + HDR, sl.h_buf.data_ptr())
        if not os.environ.get("SBM_BENCH_LATENCY_SIZING"):
            sl.ctx.set_pipeline_depth(len(slots))
        sl.run()
        fence()

    # secondary frames of the default workload: same engine, same templates, other pixels (separately timed)
    extra = {}
    if args.config == "case1" and args.frame == "scene" and not args.no_extra_frames and wl.stage == "match":
        keep = d_img.clone()
        for kind, key in (("case1", "case1_canvas"), ("tiled", "textured"), ("stagea", "stage_a")):
            fr = case1_frame(kind, ROWS, COLS)
            d_img.copy_(torch.from_numpy(np.stack([np.roll(fr, 8 * b, axis=1) for b in range(B)])).to(dev))
            for rb in ring[1:]:
                rb.copy_(d_img)
            n = max(100, min(args.steps, 300))  # secondary figures: at least 100 steps whatever --steps says
            el = timed(max(5, n // 10), n)
            k2 = kernel_pass(min(n, 30))
            fence()
            c2 = slots[0].host_counts()
            extra[key] = {"us_per_frame": el / n / B * 1e6, "steps": n,
                          "matches_frame0": int(c2[:, 0, 0].sum()),
                          "kernel_launch_us": {k: [round(x, 2) for x in v["launch_us"]] for k, v in k2.items()}}
        d_img.copy_(keep)
        for rb in ring[1:]:
            rb.copy_(d_img)
        slots[0].run()
        fence()

    if rank == 0:
        # algorithmic bytes per launch (SURVEY.md 8d / DESIGN.md), by kernel
        npx = [(ROWS >> l) * (COLS >> l) for l in range(len(T_LEVELS))]
        alg = {
            # frame read + one-hot map written (+ the next level's image, written by the fused pyrDown);
            # a launch covers the B frames of the step (refinement bytes: frame 0's figure x B)
            "k_quantize": [B * (npx[0] * (CH + 1) + npx[1] * CH), B * npx[1] * (CH + 1)],
            "k_build_lm": [B * sum(n * 9 for n in npx)],
            "k_similarity_coarse": [B * coarse_bytes],
            "k_similarity_local": [B * refine_bytes],
        }
        limiter = {
            "k_quantize": "vector-instruction issue (integer VALU at one wave-instruction per 4 cycles per SIMD); HBM traffic "
                          "equals the algorithmic bytes",
            "k_build_lm": "its stores: 2 bytes per pixel of bit strips / bit planes in 128-byte runs (20 MB in, 41 MB out per 16-frame "
                          "launch; the same launch without its level-0 stores takes 11.5 of the 16.4 us)",
            "k_similarity_coarse": "bit-plane kernel: vector issue of the bit-sliced counters and L2 -> L1 bandwidth of the items still "
                                   "alive on large template sets (c4: both near their ceilings); the longest work items' chain of "
                                   "dependent L2 round trips on a 16-frame case1 launch",
            "k_similarity_local": "bit strips (T = 4 levels): what a CU's vector memory path delivers from the L2s for dword gathers "
                                  "(~11 bytes per clock and CU: 128 bytes per feature and candidate); spread bytes (other levels): L2 line "
                                  "traffic of the 16x16 patch reads + vector issue of the response LUT",
        }
        for name in kern:
            if name in alg:
                kern[name]["algorithmic_bytes_per_step"] = float(sum(alg[name]))
                kern[name]["achieved_GBps"] = float(sum(alg[name])) / (kern[name]["ms_per_step"] * 1e-3) / 1e9
        dom = max((k for k in kern if k in alg), key=lambda k: kern[k]["ms_per_step"])
        dom_bytes = float(sum(alg[dom])) / kern[dom]["launches"]
        dom_s = kern[dom]["avg_launch_us"] * 1e-6
        achieved = dom_bytes / dom_s / 1e9
        # Counter-based figures per kernel, from the committed rocprofv3 --pmc passes of this same configuration
        # (profiles/r04_pmc.json: FETCH_SIZE / WRITE_SIZE / SQ / TCC passes, each in a run of its own, gfx950 FETCH correction
        # applied where the guide prescribes it -- tools/r04_profile.sh, tools/r04_pmc_json.py) combined with THIS run's launch
        # durations:  hbm_frac = HBM-side bytes / duration / 8 TB/s;  valu_issue_frac = vector instructions per wave x waves /
        # duration / the measured issue ceiling of this instruction mix (0.6 T wave-instructions per second: 1024 SIMDs x 2.4
        # GHz / 4.1 cycles, profiles/r02_valu_issue_rates.txt);  l2_GBps = L2 requests (128-byte lines) x 128 / duration.
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r04_pmc.json")
        pmc_key = args.config if ((args.config == "case1" and B == 16 and args.frame == "scene" and count == 360) or
                                  (args.config == "c3" and count == 3600) or args.config == "c4" or
                                  (args.config == "c5" and B == 64)) else None
        if os.path.exists(pmc) and pmc_key and world == 1:
            try:
                pj = json.load(open(pmc)).get(pmc_key, {})
                # c4's counters were collected at one rank's share (4 500 templates); every template of that configuration
                # does the same work, so the template loop's per-launch figures scale with the template count
                scale = count / 4500.0 if args.config == "c4" else 1.0
                for name in kern:
                    e = pj.get(name)
                    if not e:
                        continue
                    if scale != 1.0:
                        if not name.startswith("k_similarity"):
                            continue
                        e = dict(e, hbm_bytes_per_launch=e["hbm_bytes_per_launch"] * scale, waves=e["waves"] * scale,
                                 l2_requests=(e.get("l2_requests") or 0.0) * scale)
                    t_s = kern[name]["avg_launch_us"] * 1e-6
                    kern[name]["hbm_traffic_bytes_per_launch"] = float(e["hbm_bytes_per_launch"])
                    kern[name]["hbm_frac"] = float(e["hbm_bytes_per_launch"]) / t_s / 1e9 / HBM_PEAK_GBS
                    kern[name]["valu_issue_frac"] = float(e["valu_per_wave"]) * float(e["waves"]) / t_s / VALU_ISSUE_CEILING
                    if e.get("l2_requests"):
                        kern[name]["l2_GBps"] = float(e["l2_requests"]) * 128.0 / t_s / 1e9
                        kern[name]["l2_hit_rate"] = e.get("l2_hit_rate")
                    kern[name]["wait_inst_any_pct"] = e.get("wait_inst_any_pct")
                    kern[name]["active_valu_pct"] = e.get("active_valu_pct")
                traffic = kern.get(dom, {}).get("hbm_traffic_bytes_per_launch")
            except Exception:
                traffic = None
        total_templates = wl.n_total if wl.maps is not None else ts.n_templates
        frames_per_step_total = (wl.total_frames if wl.shard == "frames" else B)
        value = total_templates * (ROWS * COLS / 1e6) * frames_per_step_total * args.steps / elapsed
        cfg = {
            "workload": wl.desc,
            "stage": "whole Detector::match per frame" if wl.stage == "match" else "template loop (matchClass) on resident Stage-B pyramids",
            "templates_total": total_templates,
            "templates_per_gpu": count,
            "frame": [ROWS, COLS, CH],
            "parallelism": ((f"row bands of the build x{n_bands} + " if banded else "")
                            + (f"template-shard x{world}" if wl.shard == "templates" else f"frame-shard x{world}"))
                           + (" + RCCL all-gather of match lists" if collective else ""),
            "frames_per_step": frames_per_step_total,
            "frames_per_step_per_gpu": B,
            "us_per_frame": elapsed / args.steps / frames_per_step_total * 1e6,
            "frames_in_flight": len(active[0]) * B,
            "input_buffers": len(ring),
            "launch": launch,
            "exchange": exchange_path[0],
            "n_ranks_seen": n_ranks_seen,
            "hip_runtime": capi.HIP_RUNTIME,  # the process's ONE HIP runtime: "system" or torch's bundled copy (capi.py)
            ("ms_per_step_one_frame_at_a_time" if B == 1 else "ms_per_step_one_batch_at_a_time"): single_ms,
            # SURVEY 8d's two times per frame, from the per-kernel pass (kernels alone on one stream):
            # t_match = all kernels, t_templ = the template loop (coarse + refinement) only
            "t_match_kernels_us_per_frame": sum(v["ms_per_step"] for v in kern.values()) * 1e3 / B,
            "t_templ_kernels_us_per_frame": sum(v["ms_per_step"] for k, v in kern.items() if k.startswith("k_similarity")) * 1e3 / B,
            "matches_distinct": n_matches,
            "coarse_candidates_rank0": n_cand,
        }
        frac_by_frame = {}
        if strong:
            cfg["strong_estimate"] = strong
        if extra:
            # the same step on other pixels: the BASELINE configs[1] frame as the reference's demo builds it (the test image
            # on a black canvas: 65 % constant, where the gradient kernel's constant-row shortcut applies -- a best case,
            # not the headline), the image tiled over the canvas, and the scene without the object
            per_us = lambda key: total_templates * (ROWS * COLS / 1e6) / (extra[key]["us_per_frame"] * 1e-6)  # noqa: E731
            cfg["value_case1_canvas"] = per_us("case1_canvas")
            cfg["value_textured"] = per_us("textured")
            cfg["value_stage_a"] = per_us("stage_a")
            cfg["case1_canvas_us_per_frame"] = extra["case1_canvas"]["us_per_frame"]
            cfg["textured_us_per_frame"] = extra["textured"]["us_per_frame"]
            cfg["stage_a_us_per_frame"] = extra["stage_a"]["us_per_frame"]
            cfg["other_frames"] = extra
            # the dominant kernel's fraction of the HBM roofline on those frames (same algorithmic bytes, that frame's launch
            # times): round 2's headline frame was the case1 canvas
            frac_by_frame = {}
            for key, e in extra.items():
                t = e["kernel_launch_us"].get(dom)
                if t:
                    frac_by_frame[key] = dom_bytes / (sum(t) / len(t) * 1e-6) / 1e9 / HBM_PEAK_GBS
        out = {
            "metric": "templates*Mpixels/sec (whole Detector::match, frame resident in HBM)" if wl.stage == "match"
                      else "templates*Mpixels/sec (template loop only, pyramid resident)",
            "value": value,
            "unit": "templates*Mpixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "u8",
            "data": wl.data,
            "config": cfg,
            "roofline": {
                "bound": "hbm",
                "limiter": limiter.get(dom, ""),
                "kernel": dom,
                # `achieved` is algorithmic bytes / time, as the contract defines it -- EXCEPT when the dominant kernel is the
                # coarse pass: its algorithmic bytes are the reference's one byte per (template, feature, position), of which the
                # bit-plane kernel loads one or two BITS and, after the exact pruning, most not at all; that quotient is a work
                # rate (`work_rate_GBps`), not a bandwidth, so `achieved` / `frac` are then the counter-based HBM figure
                "achieved": (achieved if dom != "k_similarity_coarse" or traffic is None else traffic / dom_s / 1e9),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": (achieved if dom != "k_similarity_coarse" or traffic is None else traffic / dom_s / 1e9) / HBM_PEAK_GBS,
                "traffic": traffic,
                "work_rate_GBps": achieved,
                "valu_issue_frac": kern[dom].get("valu_issue_frac"),
                "l2_GBps": kern[dom].get("l2_GBps"),
                "algorithmic_bytes_per_launch": dom_bytes,
                "avg_launch_us": kern[dom]["avg_launch_us"],
                "frac_on_other_frames": frac_by_frame if extra else None,
                "note": "per launch = the frames of one step; figures are the mean over this kernel's launches of a step "
                        "(k_quantize: one launch per pyramid level).  `bound` names the roofline the fraction is taken against "
                        "(HBM, as BASELINE's north_star asks); `limiter` is what the counters say actually bounds the kernel",
                # the whole step against HBM, from the counters (sum over the step's launches / time of the pipelined step)
                "whole_step": ({"hbm_traffic_bytes": float(sum(v["hbm_traffic_bytes_per_launch"] * v["launches"] for v in kern.values()
                                                               if "hbm_traffic_bytes_per_launch" in v)),
                                "ms_per_step": elapsed / args.steps * 1e3,
                                "hbm_frac": float(sum(v["hbm_traffic_bytes_per_launch"] * v["launches"] for v in kern.values()
                                                      if "hbm_traffic_bytes_per_launch" in v)) / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
                                # vector instructions of all the step's launches (counted one batch at a time) over the
                                # pipelined step's time: what the overlapping batches actually saturate
                                "valu_issue_frac": float(sum(v["valu_issue_frac"] * v["avg_launch_us"] * v["launches"] for v in kern.values()
                                                             if v.get("valu_issue_frac") is not None)) * 1e-6 / (elapsed / args.steps)}
                               if traffic is not None else None),
            },
            "kernels": kern,
        }
        if world == 1 and not args.no_cpu_baseline and wl.stage == "match":
            base_ts = ts.subset(range(first, first + count))
            if base_ts.n_templates > 360:  # a bounded sample of the same workload
                base_ts = base_ts.subset(range(360))
            frame0 = np.ascontiguousarray(wl.frames[my_frames[0]])
            out["cpu_baseline"], cpu_list = cpu_baseline(base_ts, frame0, args.cpu_budget)
            if base_ts.n_templates == count:
                # the checker's list for frame 0 must be the GPU's (outside every timed region)
                if capi.canonicalize(np.ascontiguousarray(cpu_list, MATCH_DTYPE).copy()).tobytes() != capi.canonicalize(matches.copy()).tobytes():
                    raise SystemExit("GPU match list of frame 0 differs from the CPU oracle's")
                out["cpu_baseline"]["gpu_list_equals_cpu_list"] = True
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)
    for sl in slots:
        sl.ctx.close()
    if collective:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
