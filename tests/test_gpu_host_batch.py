"""-m gpu: round-3 host-side entry points of the C ABI -- the pipelined host batch (sbm_match_batch_host[_begin/_end]:
SURVEY 8f-4 "async H2D overlap"), explicit pinning (sbm_pin_host_buffer; ADVICE round 2: no implicit pinning of caller
memory), single-process multi-context matching (sbm_match_sharded, sbm_partition_templates, sbm_select_templates: SURVEY
8b).  Reference: Detector::match line2Dup.cpp:1078-1150, the OpenMP team over templates :1166-1170."""
import gc

import numpy as np
import pytest

from shape_based_matching_amd import capi, synth
from shape_based_matching_amd.templates import MATCH_DTYPE

pytestmark = pytest.mark.gpu


def key(recs):
    return sorted(np.ascontiguousarray(recs, MATCH_DTYPE).tolist())


def oracle_lists(oracle, frames, ts, thr, mask=None):
    out = []
    for fr in frames:
        p = oracle.Pyramid.build(fr, [4, 8], 30.0, mask=mask) if mask is not None else oracle.Pyramid.build(fr, [4, 8], 30.0)
        out.append(p.match(ts.levels, ts.features, ts.class_idx, ts.template_id, thr))
        p.free()
    return out


@pytest.mark.parametrize("ch", [3, 1])
def test_match_batch_host_pipelined(oracle, ctx_factory, case1, ch):
    """19 frames through sub-batches of 4 (five sub-batches: both device input buffers are re-used twice), 8 (8 + 8 + 3)
    and 32 (one), synchronous and begin / end, from pageable and from explicitly pinned memory: every frame's list is the
    oracle's"""
    ts = case1["templates"].subset(range(300, 361, 3))
    base = synth.embed(case1["test"], 640, 768, 80, 80)
    frames = [np.roll(base, 24 * b, axis=1) for b in range(19)]
    frames[7] = synth.scene_with_object(3, 640, 768, case1["test"])
    frames[11] = np.zeros_like(base)  # a frame without matches between frames with matches
    if ch == 1:
        frames = [np.ascontiguousarray(f[..., 1]) for f in frames]
    want = oracle_lists(oracle, frames, ts, 85.0)
    assert sum(len(w) for w in want) > 0 and len(want[11]) == 0
    ctx = ctx_factory()
    ctx.upload_templates(ts)
    for sub in (4, 8, 32):
        for split in (False, True):
            got = ctx.match_batch_host(frames, 85.0, cap=512, sub_batch=sub, split=split)
            assert len(got) == len(frames)
            for f in range(len(frames)):
                assert key(got[f]) == key(want[f]), (sub, split, f)
    # frames inside one explicitly pinned block
    ring = np.ascontiguousarray(np.stack(frames))
    ctx.pin_host_buffer(ring)
    with pytest.raises(capi.SbmError):
        ctx.pin_host_buffer(ring)  # already pinned by this context
    got = ctx.match_batch_host([ring[f] for f in range(len(frames))], 85.0, cap=512, sub_batch=8)
    for f in range(len(frames)):
        assert key(got[f]) == key(want[f]), f
    assert key(ctx.match(ring[3], 85.0)) == key(want[3])  # the single-frame call from the pinned block
    ctx.unpin_host_buffer(ring)
    with pytest.raises(capi.SbmError):
        ctx.unpin_host_buffer(ring)
    # per-frame capacity too small: reported, never silent
    with pytest.raises(capi.SbmError):
        ctx.match_batch_host(frames, 85.0, cap=1, sub_batch=8)
    # ... and the context stays usable
    assert key(ctx.match_batch_host(frames[:2], 85.0, cap=512)[1]) == key(want[1])


def test_match_batch_host_mask_and_geometry_change(oracle, ctx_factory, case1):
    ts = case1["templates"].subset(range(320, 361, 4))
    base = synth.embed(case1["test"], 640, 768, 80, 80)
    mask = np.zeros((640, 768), np.uint8)
    mask[100:560, 150:700] = 255
    frames = [np.roll(base, 16 * b, axis=1) for b in range(5)]
    ctx = ctx_factory()
    ctx.upload_templates(ts)
    got = ctx.match_batch_host(frames, 80.0, cap=512, sub_batch=2, mask=mask)
    for f, fr in enumerate(frames):
        p = oracle.Pyramid.build(fr, [4, 8], 30.0, mask)
        want = p.match(ts.levels, ts.features, ts.class_idx, ts.template_id, 80.0)
        p.free()
        assert key(got[f]) == key(want), f
    small = [np.ascontiguousarray(f[:512, :640]) for f in frames[:3]]  # another geometry on the same context
    want = oracle_lists(oracle, small, ts, 80.0)
    got = ctx.match_batch_host(small, 80.0, cap=512, sub_batch=2)
    for f in range(3):
        assert key(got[f]) == key(want[f]), f


def test_sbm_match_never_trusts_an_address_it_saw_before(oracle, ctx_factory, case1):
    """a caller that allocates a fresh buffer per frame.  Buffers are freed and re-allocated between calls
    (the allocator hands the same address back for the same size) with different pixels: every call must match the pixels
    it was given -- there is no implicit pinning cache keyed by (address, size) any more."""
    ts = case1["templates"].subset(range(300, 361, 4))
    base = synth.embed(case1["test"], 640, 768, 80, 80)
    ctx = ctx_factory()
    ctx.upload_templates(ts)
    variants = [base, np.ascontiguousarray(base[:, ::-1]), np.roll(base, 64, axis=1), np.zeros_like(base), np.roll(base, -40, axis=0)]
    wants = oracle_lists(oracle, variants, ts, 85.0)
    seen = set()
    for rep in range(3):
        for v, want in zip(variants, wants):
            buf = np.empty_like(base)  # fresh allocation
            buf[...] = v
            seen.add(buf.ctypes.data)
            assert key(ctx.match(buf, 85.0)) == key(want)
            del buf
            gc.collect()
    assert len(seen) < 15  # the allocator did re-use addresses: the scenario was exercised


def test_match_sharded_two_contexts_one_process(oracle, ctx_factory, case1):
    """SURVEY 8b's single-process multi-GPU entry: two contexts (both on GPU 0 here), work-balanced shards of the whole
    template list and of a two-class selection, one host thread per context, lists concatenated on the host"""
    a = case1["templates"].subset(range(300, 361, 2))
    b = case1["templates"].subset(range(0, 60, 3))
    b.class_ids = ["other"]
    from shape_based_matching_amd.templates import TemplateSet

    ts = TemplateSet.concat([a, b])
    frame = synth.embed(case1["test"], 640, 768, 80, 80)
    pyr = oracle.Pyramid.build(frame, [4, 8], 30.0)
    want_all = pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, 85.0)
    assert len(want_all) > 0
    ctxs = [ctx_factory(), ctx_factory(), ctx_factory()]
    for c in ctxs:
        c.upload_templates(ts)
    for n in (2, 3):
        parts = ctxs[0].partition_templates(640, 768, n)
        assert parts[0][0] == 0 and sum(c for _, c in parts) == ts.n_templates
        assert all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(n - 1))
        for c, (first, count) in zip(ctxs, parts):
            c.select_range(first, count)
        got = capi.Context.match_sharded(ctxs[:n], frame, 85.0)
        assert key(got) == key(want_all)
    # a selection that is not a contiguous range: templates of class "other" first, then every second one of class 0
    lst = [int(t) for t in np.nonzero(ts.class_idx == 1)[0]] + [int(t) for t in np.nonzero(ts.class_idx == 0)[0][::2]]
    parts = ctxs[0].partition_templates(640, 768, 2, lst)
    for c, (first, count) in zip(ctxs, parts):
        c.select_templates(lst[first:first + count])
    sub = ts.subset(lst)
    want = pyr.match(sub.levels, sub.features, sub.class_idx, sub.template_id, 85.0)
    got = capi.Context.match_sharded(ctxs[:2], frame, 85.0)
    assert key(got) == key(want)
    with pytest.raises(capi.SbmError):
        ctxs[0].select_templates([0, ts.n_templates])
    pyr.free()
