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
