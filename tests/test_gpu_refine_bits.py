"""-m gpu: the refinement pass on bit strips (csrc/sbm_local_bits.h) against the oracle and against the byte form.

similarityLocal (line2Dup.cpp:860-922, :986-1048; called from matchClass :1221-1293) sums response bytes {0, 3, 4} over a
16 x 16 patch; the bit form counts "response > 0" and "response == 4" bits instead (raw = 3 #any + #exact).  Both forms of
the level (sbm_set_refine_bits) must give the oracle's match list: thresholds from permissive (thousands of candidates per
frame) to strict, templates of 8 .. 8191 features (the three counter widths of the kernel), batches through the list
order, geometries whose grid is not a multiple of the builder's 16 x 64-cell tile."""
import os

import numpy as np
import pytest

from shape_based_matching_amd import capi, synth
from shape_based_matching_amd.templates import MATCH_DTYPE

pytestmark = pytest.mark.gpu
NT = min(16, os.cpu_count() or 1)


def multiset(recs):
    return sorted(np.ascontiguousarray(recs, MATCH_DTYPE).tolist())


@pytest.fixture()
def ctx_form():
    made = []

    def make(bits, **kw):
        c = capi.Context(T=kw.pop("T", (4, 8)), weak_threshold=30.0, device_id=0, max_candidates=kw.pop("max_candidates", 0))
        c.set_refine_bits(bits)
        made.append(c)
        return c

    yield make
    for c in made:
        c.close()


@pytest.mark.parametrize("bits", [True, False])
def test_case1_thresholds(oracle, ctx_form, case1, bits):
    ts = case1["templates"].subset(range(0, 360, 2))
    img = synth.embed(case1["test"], 640, 768, 40, 60)
    ctx = ctx_form(bits)
    ctx.upload_templates(ts)
    pyr = oracle.Pyramid.build(img, [4, 8], 30.0)
    sizes = []
    for thr in (40.0, 60.0, 75.0, 90.0, 97.0):
        want = pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, thr, n_threads=NT)
        got = ctx.match(img, thr)
        assert multiset(got) == multiset(want), (bits, thr)
        sizes.append(len(want))
    pyr.free()
    assert sizes[0] > 1000 and sizes[-1] > 0


def test_the_bit_form_is_what_runs(ctx_form, case1):
    """sbm_get_stats counts the bytes the refinement pass reads for frame 0: 128 per feature and candidate on bit strips,
    256 on spread bytes -- same candidates, half the bytes, so the default path is the bit form and not a silent fallback"""
    ts = case1["templates"].subset(range(0, 360, 4))
    img = synth.embed(case1["test"], 640, 768, 40, 60)
    seen = {}
    for bits in (None, True, False):
        ctx = ctx_form(bits)
        ctx.set_profiling(True)
        ctx.upload_templates(ts)
        n = len(ctx.match(img, 75.0))
        seen[bits] = (n,) + tuple(ctx.stats())
    assert seen[True][1] > 0 and seen[True][2] > 0
    assert seen[True][:2] == seen[False][:2] and 2 * seen[True][2] == seen[False][2], seen
    if os.environ.get("SBM_LOCAL_BITS", "1") != "0":
        assert seen[None] == seen[True], seen


@pytest.mark.parametrize("rows,cols", [(512, 1024), (576, 704), (480, 832), (1024, 1088)])
def test_geometries(oracle, ctx_form, case1, rows, cols):
    """grid heights / widths that are not multiples of the builder's tile (16 grid rows x 64 cells); 832 / 4 = 208 and
    1088 / 4 = 272 cells are 13 and 17 strips"""
    ts = case1["templates"].subset(range(0, 360, 5))
    img = synth.embed(case1["test"], rows, cols, 3, cols - 620)
    pyr = oracle.Pyramid.build(img, [4, 8], 30.0)
    want = multiset(pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, 70.0, n_threads=NT))
    pyr.free()
    assert len(want) > 0
    for bits in (True, False):
        ctx = ctx_form(bits)
        ctx.upload_templates(ts)
        assert multiset(ctx.match(img, 70.0)) == want, (bits, rows, cols)


def templates_from_maps(qs, nf, box, n_templates, seed):
    """templates cut out of the frame's own orientation maps (so each scores 100 where it was cut): nf[l] of the set
    pixels of level l inside the box, at a random even offset"""
    from shape_based_matching_amd.templates import from_pyramids

    rs = np.random.RandomState(seed)
    rows, cols = qs[0].shape
    pyramids, got = [], [10 ** 9] * len(qs)
    for t in range(n_templates):
        px = 64 + (rs.randint(0, cols - box - 128) // 4) * 4
        py = 64 + (rs.randint(0, rows - box - 128) // 4) * 4
        tp = []
        for l, q in enumerate(qs):
            w = box >> l
            sub = q[(py >> l) : (py >> l) + w + 1, (px >> l) : (px >> l) + w + 1]
            ys, xs = np.nonzero(sub)
            pick = rs.permutation(len(ys))[: nf[l]]
            f = np.stack([xs[pick], ys[pick], np.log2(sub[ys[pick], xs[pick]]).astype(np.int64)], axis=1)
            got[l] = min(got[l], len(pick))
            tp.append({"width": w, "height": w, "tl_x": 0, "tl_y": 0, "pyramid_level": l, "features": f})
        pyramids.append(tp)
    return from_pyramids(pyramids, "cut"), got


@pytest.mark.parametrize("nf,box", [([100, 40], 200), ([124, 60], 200), ([125, 61], 200), ([600, 200], 400), ([1020, 500], 500),
                                    ([1021, 300], 500), ([3000, 900], 700), ([8191, 4095], 840)])
def test_feature_counts(oracle, ctx_form, nf, box):
    """the kernel's counter widths change at 124 and 1020 features per template; the largest template the reference admits"""
    img = synth.scene_bgr(21, 1024, 1024, n_shapes=400)
    pyr = oracle.Pyramid.build(img, [4, 8], 30.0)
    ts, got = templates_from_maps([pyr.quantized(0), pyr.quantized(1)], nf, box, 6, 7 + nf[0])
    assert got == list(nf), (got, nf)  # the scene has enough edge pixels for the widths under test
    want = multiset(pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, 55.0, n_threads=NT))
    pyr.free()
    assert any(abs(m[2] - 100.0) < 1e-3 for m in want) and len(want) >= 6
    for bits in (True, False):
        ctx = ctx_form(bits, max_candidates=1 << 20)
        ctx.upload_templates(ts)
        got_m = multiset(ctx.match(img, 55.0))
        assert got_m == want, (bits, nf, len(got_m), len(want))


@pytest.mark.parametrize("T", [(4, 8, 8), (4, 4, 8), (8, 4, 8)])
def test_three_levels_mixed_forms(oracle, ctx_form, T):
    """three-level pyramids: a T = 4 level is refined on bit strips, a T = 8 level on spread bytes, in either order --
    the candidate records pass from one form of the pass to the other"""
    img = synth.scene_bgr(33, 1024, 1024, n_shapes=300)
    pyr = oracle.Pyramid.build(img, list(T), 30.0)
    ts, got = templates_from_maps([pyr.quantized(l) for l in range(3)], [160, 80, 40], 256, 8, 5)
    assert got == [160, 80, 40]
    want = multiset(pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, 50.0, n_threads=NT))
    pyr.free()
    assert len(want) >= 8
    for bits in (True, False):
        ctx = ctx_form(bits, T=T, max_candidates=1 << 20)
        ctx.upload_templates(ts)
        got_m = multiset(ctx.match(img, 50.0))
        assert got_m == want, (bits, T, len(got_m), len(want))


def test_batch_both_orders(oracle, ctx_form, case1):
    """6 frames through the slot order and the frame-major list order of the refinement pass, both forms"""
    import torch

    dev = torch.device("cuda", 0)
    ts = case1["templates"].subset(range(0, 360, 3))
    base = synth.embed(case1["test"], 512, 640, 0, 30)
    B = 6
    frames = np.stack([np.roll(base, 20 * b, axis=1) for b in range(B)])
    cap, rec = 8192, MATCH_DTYPE.itemsize
    d_img = torch.from_numpy(frames).to(dev)
    wants = []
    for b in range(B):
        pyr = oracle.Pyramid.build(frames[b], [4, 8], 30.0)
        wants.append(multiset(pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, 65.0, n_threads=NT)))
        pyr.free()
    assert min(len(w) for w in wants) > 50
    for bits in (True, False):
        for order in ("slots", "list"):
            ctx = ctx_form(bits)
            ctx.set_refine_order(order)
            ctx.upload_templates(ts)
            stream = torch.cuda.Stream(device=dev)
            d_out = torch.zeros(B * cap * rec, dtype=torch.uint8, device=dev)
            d_cnt = torch.zeros(B * 2, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            for _ in range(2):  # the second call replays nothing new but runs on warm buffers
                ctx.match_batch_device(d_img.data_ptr(), frames[0].size, B, 512, 640, 640 * 3, 3, 65.0, d_out.data_ptr(), cap, d_cnt.data_ptr(),
                                       stream=stream.cuda_stream)
            stream.synchronize()
            cnt = d_cnt.cpu().numpy().reshape(B, 2)
            recs = d_out.cpu().numpy().view(MATCH_DTYPE).reshape(B, cap)
            for b in range(B):
                assert cnt[b, 1] == 0
                assert multiset(recs[b, : cnt[b, 0]]) == wants[b], (bits, order, b)


def test_stage_entry_after_a_bit_strip_build(oracle, ctx_form, case1):
    """the stage entry points rebuild the response planes of a level that the last match call left as bit strips"""
    ts = case1["templates"].subset(range(0, 360, 40))
    img = synth.embed(case1["test"], 512, 640, 10, 20)
    ctx = ctx_form(True)
    ctx.upload_templates(ts)
    got = ctx.match(img, 80.0)
    assert len(got) > 0
    pyr = oracle.Pyramid.build(img, [4, 8], 30.0)
    for l in range(2):
        lm = ctx.get_linear_memories(l)
        n = (img.shape[0] >> l) * (img.shape[1] >> l)
        assert np.array_equal(lm[:, :n], pyr.lm(l)[:, :n]), l
    pyr.free()
    assert multiset(ctx.match_templates(80.0)) == multiset(got)


def test_captured_template_loop_follows_the_form_of_the_levels(oracle, ctx_form, case1):
    """graph replay of the template loop (sbm_match_templates_device with several calls in flight): a capture made on a pyramid
    that a match call built (bit strips + bit planes) must not be replayed after a stage entry point has rebuilt the levels
    as response planes at the same geometry -- the form of the levels is part of the graph's key"""
    import torch

    dev = torch.device("cuda", 0)
    ts = case1["templates"].subset(range(0, 360, 6))
    img_a = synth.embed(case1["test"], 512, 640, 10, 20)
    img_b = synth.embed(case1["test"], 512, 640, 30, 0)
    ctx = ctx_form(True)
    ctx.set_pipeline_depth(2)
    ctx.set_graph_mode(True)
    ctx.upload_templates(ts)
    cap, rec = 8192, MATCH_DTYPE.itemsize
    d_out = torch.zeros(cap * rec, dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(2, dtype=torch.int32, device=dev)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()

    def loop():
        ctx.match_templates_device(70.0, d_out.data_ptr(), cap, d_cnt.data_ptr(), stream=stream.cuda_stream)
        stream.synchronize()
        n = int(d_cnt.cpu().numpy()[0])
        return multiset(d_out.cpu().numpy().view(MATCH_DTYPE)[:n])

    want_a = multiset(ctx.match(img_a, 70.0))  # builds bit strips / bit planes
    assert len(want_a) > 0
    assert loop() == want_a and loop() == want_a  # captured, then replayed
    n_graphs = ctx.graph_count()
    pyr = oracle.Pyramid.build(img_b, [4, 8], 30.0)
    want_b = multiset(pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, 70.0, n_threads=NT))
    for l in range(2):
        ctx.set_quantized(l, pyr.quantized(l))  # a stage entry point: response planes, same geometry
    pyr.free()
    assert want_b != want_a
    assert loop() == want_b
    assert ctx.graph_count() == n_graphs + 1  # a capture of its own
    assert loop() == want_b
