"""N > 1 path on CPU: world_size-2 gloo run of the template sharding + match-list gathering logic
that bench.py / a multi-GPU host uses (RCCL takes gloo's place on the GPUs).  Each rank matches
its contiguous template shard with the CPU oracle standing in for the per-rank engine; the gathered,
canonicalised list must equal the single-process result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        from shape_based_matching_amd import sharding, synth
        from shape_based_matching_amd.templates import MATCH_DTYPE

        T = (4, 8)
        rows, cols = 384, 512
        maps, ts = synth.stage_b(4321, rows, cols, T, 48, [96, 40], templ_size=100, plant_every=4)
        # uneven work: make the second half of the templates much cheaper
        ts.levels["n_features"][24:, 1] = 8
        work = sharding.coarse_work(ts, rows, cols, T)
        parts = sharding.partition(work, world)
        assert sum(c for _, c in parts) == ts.n_templates and parts[0][0] == 0
        first, count = parts[rank]
        pyr = O.Pyramid.from_quantized(maps, T)
        shard = ts.subset(range(first, first + count))
        recs = pyr.match(shard.levels, shard.features, shard.class_idx, shard.template_id, 80.0)
        cap = 4096
        buf = np.zeros(cap, MATCH_DTYPE)
        buf[: len(recs)] = recs
        t_recs = torch.from_numpy(buf.view(np.uint8).copy())
        t_cnt = torch.tensor([len(recs), 0], dtype=torch.int32)
        gathered, counts = sharding.all_gather_matches(t_recs, t_cnt)
        assert counts.tolist()[rank] == len(recs)
        full = pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, 80.0)
        a, b = O.canonicalize(gathered), O.canonicalize(full)
        assert len(b) > 0 and a.tobytes() == b.tobytes()
        # shards are balanced by work, not by count
        w = [int(work[f : f + c].sum()) for f, c in parts]
        assert max(w) <= 1.5 * (sum(w) / world) + work.max()
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_template_shards_all_gather_world2(tmp_path, oracle):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(world))


def _frame_worker(rank, world, port, out_dir):
    """BASELINE config 5 on CPU: the frames of a batch dealt rank::world, templates replicated, one gather of the
    per-frame lists; the oracle stands in for the per-rank engine."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        from shape_based_matching_amd import sharding, synth
        from shape_based_matching_amd.templates import MATCH_DTYPE, TemplateSet

        golden = os.path.join(ROOT, "tests", "golden")
        ts = TemplateSet.load_npz(os.path.join(golden, "case1_templates.npz")).subset(range(300, 361, 5))
        img = np.load(os.path.join(golden, "case1_test_bgr.npz"))["bgr"]
        base = synth.embed(img, 640, 768, 80, 80)
        n_frames = 5  # not a multiple of the world size: the last rank slot stays empty
        frames = [np.roll(base, 24 * f, axis=1) for f in range(n_frames)]
        mine = sharding.frame_shard(n_frames, world, rank)
        assert mine.tolist() == list(range(rank, n_frames, world))
        per_rank = -(-n_frames // world)
        cap = 512
        recs = np.zeros((per_rank, cap), MATCH_DTYPE)
        counts = np.zeros((per_rank, 2), np.int32)
        for slot, f in enumerate(mine):
            pyr = O.Pyramid.build(frames[f], [4, 8], 30.0)
            got = pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, 85.0)
            pyr.free()
            recs[slot, : len(got)] = got
            counts[slot, 0] = len(got)
        lists = sharding.all_gather_frame_lists(torch.from_numpy(recs.view(np.uint8).reshape(-1).copy()),
                                                torch.from_numpy(counts.reshape(-1).copy()), n_frames)
        assert len(lists) == n_frames
        for f in range(n_frames):
            pyr = O.Pyramid.build(frames[f], [4, 8], 30.0)
            want = pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, 85.0)
            pyr.free()
            assert len(want) > 0
            assert O.canonicalize(lists[f]).tobytes() == O.canonicalize(want).tobytes(), f
        open(os.path.join(out_dir, f"fok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_frame_shards_gather_world2(tmp_path, oracle):
    world = 2
    mp.spawn(_frame_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"fok{r}") for r in range(world))


def test_partition_properties():
    from shape_based_matching_amd import sharding

    rs = np.random.RandomState(0)
    for n, k in ((360, 8), (7, 8), (1000, 3), (1, 2)):
        w = rs.randint(0, 1000, n)
        parts = sharding.partition(w, k)
        assert len(parts) == k and parts[0][0] == 0
        assert all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(k - 1))
        assert parts[-1][0] + parts[-1][1] == n
    parts = sharding.partition(np.full(360, 5), 8)
    assert all(c == 45 for _, c in parts)


def test_overflow_is_reported():
    """a rank whose list overflowed must make the gather fail loudly"""
    from shape_based_matching_amd import sharding

    port = _free_port()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        recs = torch.zeros(24 * 4, dtype=torch.uint8)
        with pytest.raises(RuntimeError):
            sharding.all_gather_matches(recs, torch.tensor([9, 0], dtype=torch.int32))
        with pytest.raises(RuntimeError):
            sharding.all_gather_matches(recs, torch.tensor([1, 1], dtype=torch.int32))
    finally:
        dist.destroy_process_group()


# ---- build sharding (round 3): row bands of the gradient stage + all-gather of the orientation maps ----------------
def _emu_band_lib():
    import ctypes as C
    import subprocess

    emu_dir = os.path.join(ROOT, "tests", "emu")
    subprocess.check_call(["make", "-s", "-C", emu_dir])
    L = C.CDLL(os.path.join(emu_dir, "libsbm_emu.so"))
    vp = C.c_void_p
    L.sbm_emu_quantize_stream_band.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_float, vp, vp, C.c_int, C.c_int, C.c_int]
    return L


def rank_band_maps(L, frame, n_levels, n_bands, band, hs=8, poison=0xAA):
    """What one rank of the build-sharded step computes for one frame: per level the rows its gradient launches write
    (sharding.band_plan) -- the kernel SOURCE on the CPU wave emulation, level l+1 reading the level image the fused
    cv::pyrDown of level l's launch produced.  Rows the rank does not compute keep the poison value."""
    from shape_based_matching_amd import sharding

    img = np.ascontiguousarray(frame)
    ch = 1 if img.ndim == 2 else 3
    plan = sharding.band_plan(img.shape[0], n_levels, n_bands, band)
    outs = []
    for l in range(n_levels):
        r, c = img.shape[:2]
        out = np.full((r, c), poison, np.uint8)
        pyr = np.full((r // 2, c // 2) + (() if ch == 1 else (3,)), poison, np.uint8)
        _, _, lo, hi = plan[l]
        rc = L.sbm_emu_quantize_stream_band(img.ctypes.data, r, c, c * ch, ch, None, 30.0, out.ctypes.data, pyr.ctypes.data, hs, lo, hi)
        assert rc == 0
        outs.append(out)
        img = pyr
    return outs, plan


def _band_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        from shape_based_matching_amd import sharding, synth

        L = _emu_band_lib()
        T = [4, 8]
        golden = os.path.join(ROOT, "tests", "golden")
        img = np.load(os.path.join(golden, "case1_test_bgr.npz"))["bgr"]
        frames = [synth.scene_with_object(7, 256, 320, img[100:260, 150:390]), synth.embed(img[:200, :300], 256, 320, 30, 10)]
        own = [[], []]
        for fr in frames:
            outs, plan = rank_band_maps(L, fr, 2, world, rank)
            for l in range(2):
                lo, hi = plan[l][:2]
                own[l].append(outs[l][lo:hi])
        assembled = [sharding.all_gather_bands(torch.from_numpy(np.stack(own[l]))).numpy() for l in range(2)]
        for f, fr in enumerate(frames):
            pyr = O.Pyramid.build(fr, T, 30.0)
            for l in range(2):
                assert np.array_equal(assembled[l][f], pyr.quantized(l)), (f, l)
            # linear memories built from the assembled maps = the single-process linear memories, bit for bit
            pb = O.Pyramid.from_quantized([assembled[0][f], assembled[1][f]], T)
            for l in range(2):
                assert np.array_equal(pb.lm(l), pyr.lm(l)), (f, l)
            pb.free()
            pyr.free()
        open(os.path.join(out_dir, f"bok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_build_bands_all_gather_world2(tmp_path, oracle):
    """rank r computes row band r of both levels' orientation maps (halo rows included), the bands are all-gathered, the
    assembled maps and the linear memories built from them equal the single-process pyramid"""
    world = 2
    mp.spawn(_band_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"bok{r}") for r in range(world))


def test_band_plan_and_band_launches_cover_exactly_what_the_next_level_reads(oracle):
    """every band count that divides the frame, 2 and 3 pyramid levels, two rows-per-wave settings: the rows each
    rank owns, taken from what its (halo-widened) launches wrote, assemble to the oracle's maps at every level"""
    from shape_based_matching_amd import sharding, synth

    assert [sharding.band_halo(3, l) for l in range(3)] == [30, 10, 0] and sharding.band_halo(1, 0) == 0
    with pytest.raises(ValueError):
        sharding.band_plan(100, 2, 8, 0)
    L = _emu_band_lib()
    fr = synth.scene_bgr(3, 192, 256)  # every level keeps cols % 4 == 0
    for n_levels, bands in ((2, (2, 4, 8)), (3, (2, 4))):
        want = [fr]
        full = []
        for l in range(n_levels):
            full.append(oracle.quantized_orientations(want[-1], 30.0)[1])
            want.append(oracle.pyrdown(want[-1]))
        for nb in bands:
            for hs in (4, 18):
                got = [np.zeros_like(m) for m in full]
                for b in range(nb):
                    outs, plan = rank_band_maps(L, fr, n_levels, nb, b, hs=hs)
                    for l in range(n_levels):
                        lo, hi = plan[l][:2]
                        got[l][lo:hi] = outs[l][lo:hi]
                for l in range(n_levels):
                    assert np.array_equal(got[l], full[l]), (n_levels, nb, hs, l)
