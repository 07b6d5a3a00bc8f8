"""-m gpu: the whole hot path through the C ABI against the oracle.

Parity contract: the set of distinct
(x, y, similarity bits, class_idx, template_id) tuples; additionally the
pre-dedup multisets must agree, which is stronger."""
import os

import numpy as np
import pytest

from shape_based_matching_amd import capi, synth
from shape_based_matching_amd.templates import MATCH_DTYPE, TemplateSet, from_pyramids

pytestmark = pytest.mark.gpu


def multiset(recs):
    r = np.ascontiguousarray(recs, MATCH_DTYPE)
    return sorted(zip(r["x"].tolist(), r["y"].tolist(), r["similarity"].view(np.uint32).tolist(), r["raw"].tolist(),
                      r["class_idx"].tolist(), r["template_id"].tolist()))


def case1_frame(case1):
    """test.cpp:341-353: pad by 250, crop to multiples of 16 -> 960 x 1088 BGR."""
    img = case1["test"]
    p = synth.embed(img, img.shape[0] + 500, img.shape[1] + 500, 250, 250)
    return np.ascontiguousarray(p[: p.shape[0] // 16 * 16, : p.shape[1] // 16 * 16])


@pytest.mark.parametrize("thr", [90.0, 60.0, 99.5])
def test_stage_b_match_templates(oracle, ctx_factory, thr):
    T = (4, 8)
    maps, ts = synth.stage_b(1234, 512, 512, T, 80, [128, 64], templ_size=120, plant_every=10)
    ctx = ctx_factory(T=T)
    ctx.upload_templates(ts)
    for l in range(2):
        ctx.set_quantized(l, maps[l])
    got = ctx.match_templates(thr)
    pyr = oracle.Pyramid.from_quantized(maps, T)
    want = pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, thr)
    assert len(want) > 0
    assert multiset(got) == multiset(want)
    assert ctx.coarse_bytes() == pyr.coarse_bytes(ts.levels, ts.features)


def test_stage_b_u8_path_and_mixed(oracle, ctx_factory):
    """< 64 features takes the reference's uint8 path at that level (:1189, :1254)"""
    T = (4, 8)
    maps, ts = synth.stage_b(99, 384, 512, T, 60, [63, 31], templ_size=100, plant_every=6)
    maps2, ts2 = synth.stage_b(98, 384, 512, T, 20, [140, 40], templ_size=100, plant_every=5)
    for l in range(2):
        maps[l] |= maps2[l]
    ts2.class_ids = ["other"]
    both = TemplateSet.concat([ts, ts2])
    ctx = ctx_factory(T=T)
    ctx.upload_templates(both)
    for l in range(2):
        ctx.set_quantized(l, maps[l])
    pyr = oracle.Pyramid.from_quantized(maps, T)
    want = pyr.match(both.levels, both.features, both.class_idx, both.template_id, 85.0)
    got = ctx.match_templates(85.0)
    assert len(want) > 0 and multiset(got) == multiset(want)
    # class selection (Detector::match class_ids, :1124-1140)
    ctx.select_classes([1])
    got1 = ctx.match_templates(85.0)
    assert multiset(got1) == multiset(want[want["class_idx"] == 1])
    # template range = one GPU's shard
    ctx.select_range(10, 30)
    gotr = ctx.match_templates(85.0)
    sel = (np.arange(both.n_templates) >= 10) & (np.arange(both.n_templates) < 40)
    keep = [i for i, r in enumerate(want) if sel[np.nonzero((both.class_idx == r["class_idx"]) & (both.template_id == r["template_id"]))[0][0]]]
    assert multiset(gotr) == multiset(want[keep])


@pytest.mark.parametrize("T", [(8,), (2, 4, 8), (5, 5)])
def test_other_pyramid_shapes(oracle, ctx_factory, T):
    L = len(T)
    rows, cols = 320, 480
    maps, ts = synth.stage_b(7 + L, rows, cols, T, 30, [96 >> l for l in range(L)], templ_size=80, plant_every=3)
    ctx = ctx_factory(T=T)
    ctx.upload_templates(ts)
    for l in range(L):
        ctx.set_quantized(l, maps[l])
    pyr = oracle.Pyramid.from_quantized(maps, T)
    for thr in (92.0, 70.0):
        want = pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, thr)
        assert multiset(ctx.match_templates(thr)) == multiset(want)
    assert len(want) > 0


def test_nonpositive_threshold_and_border_clamps(oracle, ctx_factory):
    """threshold <= 0 makes every position a candidate (score > threshold, :1208); candidates near
    the frame border exercise the clamps of :1240-1245 and best_r = best_c = -1 on empty patches"""
    T = (4, 8)
    maps, ts = synth.stage_b(5, 128, 160, T, 3, [70, 20], templ_size=40, plant_every=1, density_permille=5)
    ctx = ctx_factory(T=T)
    ctx.upload_templates(ts)
    for l in range(2):
        ctx.set_quantized(l, maps[l])
    pyr = oracle.Pyramid.from_quantized(maps, T)
    for thr in (0.0, -1.0, 10.0):
        want = pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, thr)
        got = ctx.match_templates(thr)
        assert multiset(got) == multiset(want), thr
    assert len(want) > 0


def test_full_match_case1_real_image(oracle, ctx_factory, case1):
    """reference demo angle_test (test.cpp:333-361): Detector(128,{4,8}), threshold 90, BGR frame"""
    ts = case1["templates"]
    frame = case1_frame(case1)
    assert frame.shape == (960, 1088, 3)
    ctx = ctx_factory()
    ctx.upload_templates(ts)
    got = ctx.match(frame, 90.0)
    pyr = oracle.Pyramid.build(frame, [4, 8], 30.0)
    for l in range(2):
        assert np.array_equal(ctx.get_quantized(l), pyr.quantized(l))
        assert np.array_equal(ctx.get_linear_memories(l), pyr.lm(l))
    want = pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, 90.0)
    assert len(want) > 0
    assert multiset(got) == multiset(want)
    c_got, c_want = capi.canonicalize(got), oracle.canonicalize(want)
    assert c_got.tobytes() == c_want.tobytes()
    assert int(c_got[0]["template_id"]) == 340 and abs(float(c_got[0]["similarity"]) - 98.66) < 0.01
    assert ctx.coarse_bytes() == pyr.coarse_bytes(ts.levels, ts.features)


def test_full_match_case1_gray_and_mask(oracle, ctx_factory, case1):
    ts = case1["templates"]
    frame = case1_frame(case1)
    gray = np.ascontiguousarray(frame[:, :, 1])
    mask = np.zeros(gray.shape, np.uint8)
    mask[200:800, 250:900] = 255
    ctx = ctx_factory()
    ctx.upload_templates(ts)
    for m in (None, mask):
        got = ctx.match(gray, 80.0, mask=m)
        pyr = oracle.Pyramid.build(gray, [4, 8], 30.0, mask=m)
        want = pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, 80.0)
        assert len(want) > 0 and multiset(got) == multiset(want)
        for l in range(2):
            assert np.array_equal(ctx.get_quantized(l), pyr.quantized(l))


def test_full_match_case2_real_image(oracle, ctx_factory, case2):
    """reference demo noise_test (test.cpp:450-468): Detector(30,{4,8}): uint8 similarity paths"""
    ts = case2["templates"]
    img = case2["test"]
    frame = np.ascontiguousarray(img[: img.shape[0] // 16 * 16, : img.shape[1] // 16 * 16])
    ctx = ctx_factory()
    ctx.upload_templates(ts)
    got = ctx.match(frame, 90.0)
    pyr = oracle.Pyramid.build(frame, [4, 8], 30.0)
    want = pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, 90.0)
    assert len(want) > 0 and multiset(got) == multiset(want)


def test_geometry_change_and_reuse(oracle, ctx_factory, case1):
    """one context, frames of different sizes back to back (buffers and offsets are re-derived)"""
    ts = case1["templates"].subset(range(0, 361, 30))
    ctx = ctx_factory()
    ctx.upload_templates(ts)
    frame = case1_frame(case1)
    for fr in (frame, frame[:640, :800], frame):
        fr = np.ascontiguousarray(fr)
        got = ctx.match(fr, 85.0)
        pyr = oracle.Pyramid.build(fr, [4, 8], 30.0)
        want = pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, 85.0)
        assert multiset(got) == multiset(want)


def test_capacity_overflow_is_an_error(ctx_factory):
    T = (4, 8)
    maps, ts = synth.stage_b(5, 128, 160, T, 3, [70, 20], templ_size=40, plant_every=1)
    ctx = ctx_factory(T=T, max_candidates=16)
    ctx.upload_templates(ts)
    for l in range(2):
        ctx.set_quantized(l, maps[l])
    with pytest.raises(capi.SbmError) as e:
        ctx.match_templates(0.0)
    assert e.value.code == -3


def test_full_size_properties(oracle, ctx_factory):
    """BASELINE config 3 geometry (2048 x 2048, 63/31 features) with a subset of templates: planted
    templates come back at their planted location with score 100 where no feature collided, results are
    idempotent, and the byte count matches the host-side formula."""
    T = (4, 8)
    maps, ts = synth.stage_b(31, 2048, 2048, T, 400, [63, 31], templ_size=260, plant_every=40)
    ctx = ctx_factory(T=T)
    ctx.upload_templates(ts)
    for l in range(2):
        ctx.set_quantized(l, maps[l])
    a = ctx.match_templates(90.0)
    b = ctx.match_templates(90.0)
    assert multiset(a) == multiset(b) and len(a) > 0
    best = {}
    for r in a:
        best[int(r["template_id"])] = max(best.get(int(r["template_id"]), 0.0), float(r["similarity"]))
    assert all(best.get(t, 0.0) >= 93.0 for t in range(0, 400, 40))
    pyr = oracle.Pyramid.from_quantized(maps, T)
    assert ctx.coarse_bytes() == pyr.coarse_bytes(ts.levels, ts.features)
    want = pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, 90.0, n_threads=8)
    assert multiset(a) == multiset(want)


def test_strided_rows_and_roi_views(oracle, ctx_factory, case1):
    """the C ABI takes a row stride: a frame that is a view into a wider buffer (cv::Mat ROI)"""
    import ctypes as C

    from shape_based_matching_amd.capi import _check, lib

    ts = case1["templates"].subset(range(320, 361, 4))
    frame = synth.embed(case1["test"], 640, 768, 80, 80)
    wide = np.zeros((640, 1000, 3), np.uint8)
    wide[:, 100:868] = frame
    view = wide[:, 100:868]  # not contiguous: stride 3000 bytes
    ctx = ctx_factory()
    ctx.upload_templates(ts)
    out = np.empty(4096, MATCH_DTYPE)
    n = C.c_int64(0)
    _check(lib().sbm_match(ctx._h, C.c_void_p(view.ctypes.data), 640, 768, wide.strides[0], 3, None, C.c_float(88.0),
                           out.ctypes.data_as(C.c_void_p), len(out), C.byref(n)))
    pyr = oracle.Pyramid.build(np.ascontiguousarray(frame), [4, 8], 30.0)
    want = pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, 88.0)
    assert len(want) > 0 and multiset(out[: n.value]) == multiset(want)


def test_no_templates_and_empty_selection(ctx_factory, case1):
    ctx = ctx_factory()
    frame = synth.embed(case1["test"], 640, 768, 80, 80)
    with pytest.raises(capi.SbmError) as e:  # nothing uploaded
        ctx.match(frame, 90.0)
    assert e.value.code == -4
    ts = case1["templates"].subset(range(3))
    ctx.upload_templates(ts)
    ctx.select_range(0, 0)  # an empty shard is legal: no matches
    assert len(ctx.match(frame, 90.0)) == 0


def test_full_size_gradient_stage_crop_consistency(oracle, ctx_factory):
    """BASELINE config 4 frame size (4096 x 4096): the gradient stage is local (halo 5 pixels at level 0; 2 * 5 + 2 at
    level 1 through pyrDown), so the orientation maps of the full frame must equal, window by window, the oracle's maps of
    a crop with enough margin — checked on windows that straddle tile seams, sit in the frame corners and in flat areas."""
    rows = cols = 4096
    img = synth.scene_gray(404, rows, cols, n_shapes=150)
    ctx = ctx_factory(T=(4, 8))
    ctx.build_pyramid(img)
    q0, q1 = ctx.get_quantized(0), ctx.get_quantized(1)
    assert q0.shape == (rows, cols) and q1.shape == (rows // 2, cols // 2)
    rs = np.random.RandomState(9)
    M = 32  # margin (even, > 12): the crop's own border effects stay inside it
    wins = [(0, 0), (rows - 320, cols - 320), (0, cols - 320), (1000, 2040), (2040, 1000)] + [
        (int(rs.randint(0, (rows - 320) // 2)) * 2, int(rs.randint(0, (cols - 320) // 2)) * 2) for _ in range(5)]
    nonzero = 0
    for (r0, c0) in wins:
        crop = np.ascontiguousarray(img[r0:r0 + 320, c0:c0 + 320])
        pyr = oracle.Pyramid.build(crop, [4, 8], 30.0)
        o0, o1 = pyr.quantized(0), pyr.quantized(1)
        # interior of the crop; at a frame border the crop shares the border, so no margin is needed on that side
        a0 = 0 if r0 == 0 else M
        b0 = 320 if r0 + 320 == rows else 320 - M
        a1 = 0 if c0 == 0 else M
        b1 = 320 if c0 + 320 == cols else 320 - M
        assert np.array_equal(q0[r0 + a0:r0 + b0, c0 + a1:c0 + b1], o0[a0:b0, a1:b1]), (r0, c0)
        assert np.array_equal(q1[(r0 + a0) // 2:(r0 + b0) // 2, (c0 + a1) // 2:(c0 + b1) // 2], o1[a0 // 2:b0 // 2, a1 // 2:b1 // 2]), (r0, c0)
        nonzero += int(np.count_nonzero(o0[a0:b0, a1:b1]))
        pyr.free()
    assert nonzero > 1000


def test_sbm_match_with_the_upload_in_row_bands():
    """SBM_MATCH_BANDS (off by default: slower on ROCm 7.2, see sbm_capi_match.inc): the frame crosses PCIe in 4 row bands and
    level 0's gradient tiles are launched band by band behind them; the smoke check (match list and both levels' linear
    memories against the oracle) must hold in a process that has the knob set (the tuning knobs are read once per process)"""
    import subprocess
    import sys

    from conftest import ROOT

    env = dict(os.environ, SBM_MATCH_BANDS="4")
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "smoke ok" in r.stdout, r.stdout + r.stderr
