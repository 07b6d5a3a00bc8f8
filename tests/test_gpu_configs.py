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


C3 = (2048, 3600, [63, 31], 260, 32)
C4 = (4096, 36000, [8191, 4095], 1024, 16)


@pytest.mark.parametrize("cfg,first,count,n_parts", [(C4, 4500, 4500, 4), (C4, 0, 36000, 8), (C3, 0, 3600, 8)],
                         ids=["c4-one-rank-share", "c4-all-36000", "c3-all-3600"])
def test_config3_config4_full_size_properties(ctx_factory, cfg, first, count, n_parts):
    """BASELINE config 4 at the size ONE rank of the 8-GPU job runs it -- 4 500 of the 36 000 templates x 8191 / 4095 features
    (templates 4500 .. 8999: 55 M features) -- and at its FULL stated size, all 36 000 templates on one GPU (442 M features:
    5.3 GB of generated template data on the host, 6.6 GB in HBM), on 4096^2 maps -- far beyond what the oracle finishes in
    seconds (and BASELINE config 3 at its full size, 2048^2 x 3600 templates x 63 / 31 features, the uint8 paths), so the
    check is through properties the domain offers: (1) the planted templates of the range, and nothing else, come back
    (synth.stage_b_fixed plants template k * n_total / n_plants; features that share a pixel keep the score just below 100); (2) the match list of the
    whole range is the union of the lists of any partition of it (sbm_partition_templates, sbm_select_range): what template
    sharding over GPUs relies on; (3) repeating the call gives the same list.  test_config4_thousand_template_shard_of_the_
    full_set pins a 1024-template slice of the same set to the oracle."""
    T = (4, 8)
    side, n_total, nf, box, n_plants = cfg
    maps, ts = synth.stage_b_fixed(1234, side, side, T, n_total, nf, templ_size=box, n_plants=n_plants, first=first, count=count)
    assert ts.n_templates == count and int(ts.template_id[0]) == first
    ctx = ctx_factory(T=T, max_candidates=1 << 22)
    ctx.upload_templates(ts)
    for l in range(2):
        ctx.set_quantized(l, maps[l])
    whole = multiset(ctx.match_templates(90.0))
    assert whole == multiset(ctx.match_templates(90.0))
    planted = {k * n_total // n_plants for k in range(n_plants)} & set(range(first, first + count))
    assert len(planted) == n_plants * count // n_total
    assert {r[5] for r in whole} == planted and all(r[2] > 90.0 for r in whole)
    parts = ctx.partition_templates(side, side, n_parts)
    assert sum(c for _, c in parts) == count and parts[0][0] == 0 and len(parts) == n_parts
    union = []
    for f, c in parts:
        ctx.select_range(f, c)
        union += multiset(ctx.match_templates(90.0))
    assert sorted(union) == whole


@pytest.mark.parametrize("cfg,first,count", [(C3, 0, 3600), (C4, 4500, 4500), (C4, 0, 36000)],
                         ids=["c3-all-3600", "c4-one-rank-share-4500", "c4-all-36000"])
def test_config3_full_and_config4_rank_share_against_the_oracle(oracle, ctx_factory, cfg, first, count):
    """Round 4 (the full-size checks were self-comparisons): BASELINE config 3 at its FULL stated
    size -- all 3600 templates on 2048^2 maps -- and config 4 at the size one rank of the 8-GPU job runs it -- 4 500 templates
    x 8191 / 4095 features on 4096^2 maps --, and config 4 at its FULL stated size, all 36 000 templates (442 M features),
    against the ORACLE's template loop on the box's host cores (the 16-thread oracle needs a few seconds for the first two,
    under a minute for the last): match multiset and the coarse pass's algorithmic byte count, bit for bit.  Inputs are
    the bench's (synth.stage_b_fixed, seed 1234)."""
    T = (4, 8)
    side, n_total, nf, box, n_plants = cfg
    maps, ts = synth.stage_b_fixed(1234, side, side, T, n_total, nf, templ_size=box, n_plants=n_plants, first=first, count=count)
    ctx = ctx_factory(T=T, max_candidates=1 << 22)
    ctx.upload_templates(ts)
    for l in range(2):
        ctx.set_quantized(l, maps[l])
    got = ctx.match_templates(90.0)
    pyr = oracle.Pyramid.from_quantized(maps, T)
    want = pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, 90.0, n_threads=min(16, os.cpu_count() or 1))
    assert len(want) > 0
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


def test_config5_full_size_properties(ctx_factory, case1):
    """BASELINE config 5 at its stated size: 64 frames of 1920 x 1072 x 3 (the tiled case1 image, shifted per frame, as
    bench.py --config c5 makes them) x 1000 templates in ONE call -- the size at which the refinement pass walks the
    candidates as one frame-major list (the batch's planes exceed the L2s).  Properties: every frame's list equals the
    list the same frame gets in a 16-frame call (refinement by per-frame slots, other launch sizes) and, for a few
    frames, the single-frame entry point's; no list overflows; frames 0 and 2 (shifted by 16 pixels = one period of every
    stride of the pyramid) give the same number of matches away from the left and right borders."""
    import torch

    from shape_based_matching_amd.templates import TemplateSet

    dev = torch.device("cuda", 0)
    base = case1["templates"]
    sets = []
    for k, take in enumerate((360, 360, 280)):
        s = base.subset(range(take))
        s.class_ids = [f"test{k}"]
        sets.append(s)
    ts = TemplateSet.concat(sets)
    assert ts.n_templates == 1000
    rows, cols, B = 1072, 1920, 64
    img = case1["test"]
    reps = (-(-rows // img.shape[0]), -(-cols // img.shape[1]), 1)
    frame = np.ascontiguousarray(np.tile(img, reps)[:rows, :cols])
    frames = np.stack([np.roll(frame, 8 * b, axis=1) for b in range(B)])
    cap, rec, thr = 2048, MATCH_DTYPE.itemsize, 90.0
    fs = rows * cols * 3
    ctx = ctx_factory()
    ctx.upload_templates(ts)
    stream = torch.cuda.Stream(device=dev)
    d_imgs = torch.from_numpy(frames).to(dev)
    d_out = torch.zeros(B * cap * rec, dtype=torch.uint8, device=dev)
    d_cnt = torch.full((2 * B,), -1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.match_batch_device(d_imgs.data_ptr(), fs, B, rows, cols, cols * 3, 3, thr, d_out.data_ptr(), cap, d_cnt.data_ptr(),
                           stream=stream.cuda_stream)
    stream.synchronize()
    cnt = d_cnt.cpu().numpy().reshape(B, 2).copy()
    out = d_out.cpu().numpy().reshape(B, cap * rec).copy()
    assert (cnt[:, 1] == 0).all() and (cnt[:, 0] > 0).all() and (cnt[:, 0] <= cap).all()
    whole = [key(out[b].view(MATCH_DTYPE)[: cnt[b, 0]]) for b in range(B)]
    # the same frames, 16 per call
    for g in range(0, B, 16):
        d_cnt.fill_(-1)
        torch.cuda.synchronize()
        ctx.match_batch_device(d_imgs.data_ptr() + g * fs, fs, 16, rows, cols, cols * 3, 3, thr, d_out.data_ptr(), cap, d_cnt.data_ptr(),
                               stream=stream.cuda_stream)
        stream.synchronize()
        c16 = d_cnt.cpu().numpy().reshape(B, 2)[:16]
        o16 = d_out.cpu().numpy().reshape(B, cap * rec)[:16]
        for b in range(16):
            assert c16[b, 1] == 0 and key(o16[b].view(MATCH_DTYPE)[: c16[b, 0]]) == whole[g + b], g + b
    one_out = torch.zeros(cap * rec, dtype=torch.uint8, device=dev)
    one_cnt = torch.zeros(2, dtype=torch.int32, device=dev)
    for b in (0, 37, 63):
        ctx.match_device(d_imgs.data_ptr() + b * fs, rows, cols, cols * 3, 3, thr, one_out.data_ptr(), cap, one_cnt.data_ptr(),
                         stream=stream.cuda_stream)
        stream.synchronize()
        n = int(one_cnt.cpu().numpy()[0])
        assert key(one_out.cpu().numpy().view(MATCH_DTYPE)[:n]) == whole[b], b
    # a shift by 16 pixels moves every interior match by 16 pixels
    def interior(recs, shift):
        return sorted((r[0] - shift, r[1], r[3], r[4], r[5]) for r in recs if 400 + shift <= r[0] < cols - 700 + shift)

    assert interior(whole[0], 0) == interior(whole[2], 16) and len(interior(whole[0], 0)) > 0
    # Round 4: and against the ORACLE, every one of the 64 frames x 1000 templates -- the whole
    # configuration at its stated size, not a sample (the 16-thread oracle needs about 0.1 s per frame)
    from oracle import oracle as O

    for b in range(B):
        pyr = O.Pyramid.build(frames[b], [4, 8], 30.0)
        want = pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, thr, n_threads=min(16, os.cpu_count() or 1))
        pyr.free()
        assert key(want) == whole[b], b


def key(recs):
    return sorted(np.ascontiguousarray(recs, MATCH_DTYPE).tolist())
