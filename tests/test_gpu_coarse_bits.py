"""-m gpu: the coarsest level's BIT planes (round 4) -- what the coarse pass reads instead of the reference's one byte per
(position, orientation) (similarity, line2Dup.cpp:843-856; SIMILARITY_LUT's values {0, 3, 4}, :632-635).

Three producers must give the same planes, both equal to the oracle's byte linear memories packed by numpy
(plane o: LM[o] > 0, plane 8 + o: LM[o] == 4, flat order, zero tail): the fused one inside the one-launch builder of the
match entry points (k_build_lm_rows, compact == 3), the spread-plane pack of the grids that one does not take
(k_pack_bitplanes_spread) and the generic byte -> bit pack the stage entry points use.  The
match lists through either must equal the oracle's -- also after the form of the coarsest level changed under a
context (bits only -> response planes on demand -> bits again)."""
import os

import numpy as np
import pytest

from shape_based_matching_amd import capi, synth
from shape_based_matching_amd.templates import MATCH_DTYPE, from_pyramids

pytestmark = pytest.mark.gpu


def multiset(recs):
    return sorted(np.ascontiguousarray(recs, MATCH_DTYPE).tolist())


def packed(lm):
    """[8][stride] response bytes -> [16][stride / 8] bit planes, little-endian bit order"""
    return np.concatenate([np.packbits(lm > 0, axis=1, bitorder="little"), np.packbits(lm == 4, axis=1, bitorder="little")])


@pytest.mark.parametrize("shape,T,ch", [((512, 1024), (4, 8), 3), ((1024, 1024), (4, 8), 1), ((512, 1024), (8, 8), 3), ((512, 1024), (4,), 1),
                                        ((512, 640), (4, 8), 3)])
def test_fused_bit_planes_equal_packed_oracle_planes(oracle, ctx_factory, case1, shape, T, ch):
    """sbm_match builds the coarsest level as bit planes only (W * H % 256 == 0) or as one plane of spread bytes that
    k_pack_bitplanes_spread turns into them ((512, 640): W * H = 1280 is not a multiple of 256); either way they are the
    oracle's planes, and the response planes the stage accessor then asks for are rebuilt from the orientation map /
    expanded from the spread plane"""
    rows, cols = shape
    ts = case1["templates"].subset(range(0, 360, 9))
    if len(T) == 1:  # a one-level pyramid: the templates' level 0 alone (the coarse pass then emits the matches itself)
        ts = from_pyramids([[{"width": int(lv["width"]), "height": int(lv["height"]),
                              "features": [(int(f["x"]), int(f["y"]), int(f["label"])) for f in ts.feats_of(t, 0)]}]
                            for t, lv in enumerate(ts.levels[:, 0])])
    img = synth.embed(case1["test"] if ch == 3 else case1["test"][..., 1].copy(), rows, cols, 10, 20)
    ctx = ctx_factory(T=T)
    ctx.upload_templates(ts)
    got = ctx.match(img, 85.0)
    pyr = oracle.Pyramid.build(img, list(T), 30.0)
    want = pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, 85.0, n_threads=min(16, os.cpu_count() or 1))
    assert multiset(got) == multiset(want)
    lc = len(T) - 1
    bits = ctx.get_coarse_bitplanes(0)
    assert np.array_equal(bits, packed(pyr.lm(lc)))
    assert np.array_equal(ctx.get_linear_memories(lc), pyr.lm(lc))  # response planes on demand
    got2 = ctx.match(img, 85.0)  # and the bit planes again
    assert multiset(got2) == multiset(want)
    pyr.free()


def test_bit_planes_of_a_batch_and_byte_kernels_after_them(oracle, ctx_factory, case1):
    """every frame of a batch has its own planes; a byte kernel chosen afterwards (sbm_set_coarse_mode) builds its response
    planes again; thresholds < 0 (every position a candidate: byte kernels) after a bits-only build"""
    import torch

    dev = torch.device("cuda", 0)
    ts = case1["templates"].subset(range(0, 360, 5))
    base = synth.embed(case1["test"], 512, 1024, 0, 0)
    B = 5
    frames = np.stack([np.roll(base, 32 * b, axis=0) for b in range(B)])
    cap, rec = 4096, MATCH_DTYPE.itemsize
    d_img = torch.from_numpy(frames).to(dev)
    ctx = ctx_factory()
    ctx.upload_templates(ts)
    stream = torch.cuda.Stream(device=dev)
    d_out = torch.zeros(B * cap * rec, dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(B * 2, dtype=torch.int32, device=dev)
    pyrs = [oracle.Pyramid.build(frames[b], [4, 8], 30.0) for b in range(B)]
    wants = [multiset(p.match(ts.levels, ts.features, ts.class_idx, ts.template_id, 88.0, n_threads=min(16, os.cpu_count() or 1))) for p in pyrs]
    for mode in ("auto", "wave", "bits"):
        ctx.set_coarse_mode(mode)
        torch.cuda.synchronize()
        ctx.match_batch_device(d_img.data_ptr(), frames[0].size, B, 512, 1024, 1024 * 3, 3, 88.0, d_out.data_ptr(), cap, d_cnt.data_ptr(),
                               stream=stream.cuda_stream)
        stream.synchronize()
        cnt = d_cnt.cpu().numpy().reshape(B, 2)
        recs = d_out.cpu().numpy().view(MATCH_DTYPE).reshape(B, cap)
        for b in range(B):
            assert cnt[b, 1] == 0
            assert multiset(recs[b, : cnt[b, 0]]) == wants[b], (mode, b)
            if mode != "wave":
                assert np.array_equal(ctx.get_coarse_bitplanes(b), packed(pyrs[b].lm(1))), (mode, b)
    # single-frame entry point, then the template loop alone with a negative threshold on what it left resident
    ctx.set_coarse_mode("auto")
    small = ts.subset(range(0, ts.n_templates, 24))
    ctx.upload_templates(small)
    got = ctx.match(frames[2], 88.0)
    want = pyrs[2].match(small.levels, small.features, small.class_idx, small.template_id, 88.0)
    assert multiset(got) == multiset(want)
    for p in pyrs:
        p.free()


def test_stage_path_packs_bit_planes_from_response_planes(oracle, ctx_factory):
    """sbm_set_quantized + sbm_match_templates (BASELINE configs 3 and 4 run this way): the generic pack, also for a grid
    whose W * H is not a multiple of 256 and for T = 5 (no register-only builder)"""
    for rows, cols, T, nf in ((480, 608, (4, 8), [40, 20]), (400, 400, (5, 5), [30, 16]), (1024, 1024, (4, 8), [200, 100])):
        maps, ts = synth.stage_b(11, rows, cols, T, 60, nf, templ_size=120, plant_every=7)
        ctx = ctx_factory(T=T, max_candidates=1 << 20)
        ctx.upload_templates(ts)
        for l in range(2):
            ctx.set_quantized(l, maps[l])
        pyr = oracle.Pyramid.from_quantized(maps, T)
        for thr in (92.0, 70.0):
            got = ctx.match_templates(thr)
            want = pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, thr, n_threads=min(16, os.cpu_count() or 1))
            assert len(want) > 0
            assert multiset(got) == multiset(want), (rows, T, thr)
        assert np.array_equal(ctx.get_coarse_bitplanes(0), packed(pyr.lm(1)))
        pyr.free()
