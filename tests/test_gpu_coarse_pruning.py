"""-m gpu: the coarse pass's exact pruning and its three kernels (bytes: four waves per item / one wave per item; bit planes).

The coarse pass (similarity, line2Dup.cpp:807-858 / :924-984, scanned at :1199-1216) stops an item after a prefix of
its features when no position can reach the threshold any more.  Nothing that could have reached the threshold may be
dropped: over a range of thresholds -- which moves the prefix from "no pruning" (low thresholds) to 8 features -- and
with either kernel forced (sbm_set_coarse_mode), the match lists must equal the oracle's."""
import os

import numpy as np
import pytest

from shape_based_matching_amd import capi, synth
from shape_based_matching_amd.templates import MATCH_DTYPE

pytestmark = pytest.mark.gpu


def multiset(recs):
    return sorted(np.ascontiguousarray(recs, MATCH_DTYPE).tolist())


@pytest.fixture()
def forced_ctx():
    made = []

    def make(mode, **kw):
        c = capi.Context(T=kw.pop("T", (4, 8)), weak_threshold=30.0, device_id=0, max_candidates=kw.pop("max_candidates", 0))
        c.set_coarse_mode(mode)
        made.append(c)
        return c

    yield make
    for c in made:
        c.close()


@pytest.mark.parametrize("mode", ["block", "wave", "bits"])
def test_case1_thresholds(oracle, forced_ctx, case1, mode):
    """131 / 71 features: thresholds 50 (no prefix short enough: unpruned) .. 99 (prefix of 8 features)"""
    ts = case1["templates"].subset(range(0, 360, 3))
    img = synth.embed(case1["test"], 640, 768, 40, 60)
    ctx = forced_ctx(mode)
    ctx.upload_templates(ts)
    pyr = oracle.Pyramid.build(img, [4, 8], 30.0)
    n_lists = []
    for thr in (50.0, 70.0, 80.0, 86.0, 90.0, 95.0, 99.0):
        want = pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, thr, n_threads=min(16, os.cpu_count() or 1))
        got = ctx.match(img, thr)
        assert multiset(got) == multiset(want), (mode, thr)
        n_lists.append(len(want))
    pyr.free()
    assert n_lists[0] > n_lists[4] > 0


@pytest.mark.parametrize("mode", ["block", "wave", "bits"])
def test_stage_b_small_and_huge_templates(oracle, forced_ctx, mode):
    """31 features at the coarse level (uint8 path, prefix 8) and 4095 (prefix 1024; the planted templates' items run
    the remaining 3071 features)"""
    T = (4, 8)
    for rows, nt, nf, box, plant, thr in ((1024, 120, [63, 31], 260, 10, 90.0), (2048, 12, [8191, 4095], 1024, 4, 85.0)):
        maps, ts = synth.stage_b(77, rows, rows, T, nt, nf, templ_size=box, plant_every=plant)
        ctx = forced_ctx(mode, max_candidates=1 << 22)
        ctx.upload_templates(ts)
        for l in range(2):
            ctx.set_quantized(l, maps[l])
        got = ctx.match_templates(thr)
        pyr = oracle.Pyramid.from_quantized(maps, T)
        want = pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, thr, n_threads=min(16, os.cpu_count() or 1))
        pyr.free()
        assert len(want) > 0
        assert multiset(got) == multiset(want), (mode, nf)


def test_batch_both_kernels_agree(oracle, forced_ctx, case1):
    """a batch of 8 frames (frame -> XCD mapping active) through both kernels; every frame's list against the oracle"""
    import torch

    dev = torch.device("cuda", 0)
    ts = case1["templates"]
    base = synth.embed(case1["test"], 512, 640, 0, 30)
    B = 8
    frames = np.stack([np.roll(base, 16 * b, axis=1) for b in range(B)])
    cap, rec = 4096, MATCH_DTYPE.itemsize
    d_img = torch.from_numpy(frames).to(dev)
    wants = []
    for b in range(B):
        pyr = oracle.Pyramid.build(frames[b], [4, 8], 30.0)
        wants.append(multiset(pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, 88.0, n_threads=min(16, os.cpu_count() or 1))))
        pyr.free()
    for mode in ("block", "wave", "bits"):
        ctx = forced_ctx(mode)
        ctx.upload_templates(ts)
        stream = torch.cuda.Stream(device=dev)
        d_out = torch.zeros(B * cap * rec, dtype=torch.uint8, device=dev)
        d_cnt = torch.zeros(B * 2, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()  # the fills run on torch's stream, the match on `stream`
        ctx.match_batch_device(d_img.data_ptr(), frames[0].size, B, 512, 640, 640 * 3, 3, 88.0, d_out.data_ptr(), cap, d_cnt.data_ptr(),
                               stream=stream.cuda_stream)
        stream.synchronize()
        cnt = d_cnt.cpu().numpy().reshape(B, 2)
        recs = d_out.cpu().numpy().view(MATCH_DTYPE).reshape(B, cap)
        for b in range(B):
            assert cnt[b, 1] == 0
            assert multiset(recs[b, : cnt[b, 0]]) == wants[b], (mode, b)


@pytest.mark.parametrize("mode", ["wave", "bits"])
@pytest.mark.parametrize("n_templates,B", [(7, 3), (121, 5), (358, 2)])
def test_wave_kernel_ragged_template_count_and_batch(oracle, forced_ctx, case1, n_templates, B, mode):
    """one wave per item, four template slots per workgroup: template counts that are not multiples of 4, batches
    without the frame -> XCD mapping (B not a multiple of 8), a frame geometry whose chunk count is not a multiple of 8"""
    import torch

    dev = torch.device("cuda", 0)
    idx = sorted(set(np.linspace(0, 359, n_templates).astype(int).tolist()))  # spread over all rotations
    if len(idx) % 4 == 0:
        idx = idx[:-1]
    ts = case1["templates"].subset(idx)
    base = synth.embed(case1["test"], 576, 704, 20, 10)
    frames = np.stack([np.roll(base, 24 * b, axis=1) for b in range(B)])
    cap, rec = 4096, MATCH_DTYPE.itemsize
    d_img = torch.from_numpy(frames).to(dev)
    ctx = forced_ctx(mode)
    ctx.upload_templates(ts)
    stream = torch.cuda.Stream(device=dev)
    d_out = torch.zeros(B * cap * rec, dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(B * 2, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()  # the fills run on torch's stream, the match on `stream`
    ctx.match_batch_device(d_img.data_ptr(), frames[0].size, B, 576, 704, 704 * 3, 3, 85.0, d_out.data_ptr(), cap, d_cnt.data_ptr(),
                           stream=stream.cuda_stream)
    stream.synchronize()
    cnt = d_cnt.cpu().numpy().reshape(B, 2)
    recs = d_out.cpu().numpy().view(MATCH_DTYPE).reshape(B, cap)
    total = 0
    for b in range(B):
        pyr = oracle.Pyramid.build(frames[b], [4, 8], 30.0)
        want = multiset(pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, 85.0, n_threads=min(16, os.cpu_count() or 1)))
        pyr.free()
        assert cnt[b, 1] == 0
        assert multiset(recs[b, : cnt[b, 0]]) == want, (n_templates, b)
        total += len(want)
    assert total > 0 or n_templates < 20
    assert ts.n_templates % 4 != 0
