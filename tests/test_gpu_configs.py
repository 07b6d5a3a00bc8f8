"""-m gpu: the BASELINE.json configurations at their FULL frame geometry through the C ABI against the oracle.

configs[3] (config 4): 4096 x 4096, templates of 8191 / 4095 features (the int16 maximum, line2Dup.cpp:811, :1195),
template box 1024 -- a shard of the 36 000-template set small enough for the oracle to finish in seconds.
configs[4] (config 5): one 1920 x 1072 frame of the stream with all 1000 templates (128 / 64 features).
configs[2] (config 3): 2048 x 2048, 63 / 31 features (the uint8 paths, :924-984, :986-1048), 400 of the 3600 templates.
Reference functions: similarity / similarity_64 :807-858 / :924-984, similarityLocal(_64) :860-922 / :986-1048,
matchClass :1160-1297."""
import os

import numpy as np
import pytest

from shape_based_matching_amd import synth
from shape_based_matching_amd.templates import MATCH_DTYPE

pytestmark = pytest.mark.gpu


def multiset(recs):
    return sorted(np.ascontiguousarray(recs, MATCH_DTYPE).tolist())


def run_stage_b(oracle, ctx_factory, rows, cols, nt, nf, box, plant_every, thr=90.0, seed=1234):
    T = (4, 8)
    maps, ts = synth.stage_b(seed, rows, cols, T, nt, nf, templ_size=box, plant_every=plant_every)
    ctx = ctx_factory(T=T, max_candidates=1 << 22)
    ctx.upload_templates(ts)
    for l in range(2):
        ctx.set_quantized(l, maps[l])
    got = ctx.match_templates(thr)
    pyr = oracle.Pyramid.from_quantized(maps, T)
    want = pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, thr, n_threads=min(16, os.cpu_count() or 1))
    assert len(want) > 0
    assert multiset(got) == multiset(want)
    assert ctx.coarse_bytes() == pyr.coarse_bytes(ts.levels, ts.features)
    pyr.free()
    return len(want)


def test_config4_full_geometry_max_features(oracle, ctx_factory):
    """4096^2 linear memories (128 MiB + 32 MiB), 24 templates x 8191 / 4095 features, box 1024"""
    run_stage_b(oracle, ctx_factory, 4096, 4096, 24, [8191, 4095], 1024, plant_every=6)


def test_config4_thousand_template_shard_of_the_full_set(oracle, ctx_factory):
    """A 1024-template shard (templates 4000 .. 5023, which hold the plant at 4500) of BASELINE config 4's full 36 000
    x 8191 / 4095-feature set on 4096^2 maps, generated the way bench.py --config c4 generates a rank's shard
    (synth.stage_b_fixed: the number of planted templates does not grow with the template count, so the maps keep their
    sparse density and the candidate list stays small at any shard size)."""
    T = (4, 8)
    maps, ts = synth.stage_b_fixed(1234, 4096, 4096, T, 36000, [8191, 4095], templ_size=1024, n_plants=16, first=4000, count=1024)
    assert ts.template_id[0] == 4000 and ts.n_templates == 1024
    ctx = ctx_factory(T=T, max_candidates=1 << 22)
    ctx.upload_templates(ts)
    for l in range(2):
        ctx.set_quantized(l, maps[l])
    got = ctx.match_templates(90.0)
    pyr = oracle.Pyramid.from_quantized(maps, T)
    want = pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, 90.0, n_threads=min(16, os.cpu_count() or 1))
    assert len(want) > 0 and {int(r[5]) for r in np.ascontiguousarray(want, MATCH_DTYPE).tolist()} == {4500}
    assert multiset(got) == multiset(want)
    assert ctx.coarse_bytes() == pyr.coarse_bytes(ts.levels, ts.features)
    pyr.free()


def test_config5_one_frame_all_templates(oracle, ctx_factory):
    """1920 x 1072 (the 1080p frame cropped to multiples of 16, test.cpp:349-353), 1000 templates x 128 / 64"""
    run_stage_b(oracle, ctx_factory, 1072, 1920, 1000, [128, 64], 260, plant_every=40)


def test_config3_uint8_paths(oracle, ctx_factory):
    """2048^2, 63 / 31 features: similarity_64 / similarityLocal_64"""
    run_stage_b(oracle, ctx_factory, 2048, 2048, 400, [63, 31], 260, plant_every=40)


def test_set_quantized_then_batch_at_new_geometry(oracle, ctx_factory, case1):
    """A context that matched at geometry A, then received every level through sbm_set_quantized at a LARGER
    geometry B, then runs a batch at B with the same channel count and no more frames than before: the per-level
    buffers must be re-sized for the batch (one-frame buffers left by set_quantized would be overrun)."""
    import torch

    dev = torch.device("cuda", 0)
    ts = case1["templates"].subset(range(300, 361, 4))
    img = case1["test"]
    small = synth.embed(img[:300, :300], 320, 320, 10, 10)
    big = synth.embed(img, 640, 768, 80, 80)
    ctx = ctx_factory()
    ctx.upload_templates(ts)
    cap, rec = 1024, MATCH_DTYPE.itemsize
    stream = torch.cuda.Stream(device=dev)
    B = 3
    d_small = torch.from_numpy(np.stack([small] * B)).to(dev)
    d_out = torch.zeros(B * cap * rec, dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(B * 2, dtype=torch.int32, device=dev)
    ctx.match_batch_device(d_small.data_ptr(), small.size, B, 320, 320, 320 * 3, 3, 80.0, d_out.data_ptr(), cap, d_cnt.data_ptr(),
                           stream=stream.cuda_stream)
    stream.synchronize()
    pyr = oracle.Pyramid.build(big, [4, 8], 30.0)
    for l in range(2):
        ctx.set_quantized(l, pyr.quantized(l))
    frames = [big, np.ascontiguousarray(big[:, ::-1]), np.roll(big, 32, axis=1)]
    d_big = torch.from_numpy(np.stack(frames)).to(dev)
    d_cnt.fill_(-1)
    torch.cuda.synchronize()  # uploads and fills ran on torch's stream, the match runs on `stream`
    ctx.match_batch_device(d_big.data_ptr(), big.size, B, 640, 768, 768 * 3, 3, 80.0, d_out.data_ptr(), cap, d_cnt.data_ptr(),
                           stream=stream.cuda_stream)
    stream.synchronize()
    cnt = d_cnt.cpu().numpy().reshape(-1, 2)
    out = d_out.cpu().numpy().reshape(B, cap * rec)
    for f, fr in enumerate(frames):
        p = oracle.Pyramid.build(fr, [4, 8], 30.0)
        want = p.match(ts.levels, ts.features, ts.class_idx, ts.template_id, 80.0)
        p.free()
        assert cnt[f, 1] == 0 and cnt[f, 0] == len(want)
        assert multiset(out[f].view(MATCH_DTYPE)[: cnt[f, 0]]) == multiset(want)
    assert len(want) > 0
    pyr.free()
