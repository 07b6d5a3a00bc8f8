"""-m gpu: sbm_match_device (frame resident in HBM, caller's stream, caller's result buffers), in
hipGraph mode and plain-stream mode, with and without the pinned-host result mirror."""
import numpy as np
import pytest

from shape_based_matching_amd import capi, synth
from shape_based_matching_amd.templates import MATCH_DTYPE

pytestmark = pytest.mark.gpu


def key(recs):
    return sorted(np.ascontiguousarray(recs, MATCH_DTYPE).tolist())


def frame_of(case1):
    img = case1["test"]
    return synth.embed(img, 640, 768, 80, 80)


@pytest.mark.parametrize("graph", [True, False])
def test_match_device_matches_oracle(oracle, ctx_factory, case1, graph):
    import torch

    dev = torch.device("cuda", 0)
    ts = case1["templates"].subset(range(280, 361, 2))
    frame = frame_of(case1)
    ctx = ctx_factory()
    ctx.set_graph_mode(graph)
    ctx.upload_templates(ts)
    cap = 2048
    stream = torch.cuda.Stream(device=dev)
    d_imgs = [torch.from_numpy(frame).to(dev), torch.from_numpy(np.ascontiguousarray(frame[:, ::-1])).to(dev)]
    d_out = torch.zeros(cap * MATCH_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(2, dtype=torch.int32, device=dev)
    h_out = torch.zeros(cap * MATCH_DTYPE.itemsize, dtype=torch.uint8).pin_memory()
    h_cnt = torch.zeros(2, dtype=torch.int32).pin_memory()
    frames = [frame, np.ascontiguousarray(frame[:, ::-1])]
    want = {}
    for thr in (90.0, 75.0):
        for i, fr in enumerate(frames):
            pyr = oracle.Pyramid.build(fr, [4, 8], 30.0)
            want[(thr, i)] = pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, thr)
    assert len(want[(90.0, 0)]) > 0
    for mirror in (False, True):
        ctx.set_result_mirror(h_out.data_ptr() if mirror else 0, h_cnt.data_ptr() if mirror else 0)
        for rep in range(3):  # replays of the captured graphs, alternating frames and thresholds
            for thr in (90.0, 75.0):
                for i in range(2):
                    h_cnt.zero_()
                    with torch.cuda.stream(stream):
                        ctx.match_device(d_imgs[i].data_ptr(), frame.shape[0], frame.shape[1], frame.shape[1] * 3, 3, thr,
                                         d_out.data_ptr(), cap, d_cnt.data_ptr(), stream=stream.cuda_stream)
                    stream.synchronize()
                    cnt = d_cnt.cpu().numpy()
                    assert cnt[1] == 0
                    got = d_out.cpu().numpy().view(MATCH_DTYPE)[: cnt[0]]
                    assert key(got) == key(want[(thr, i)]), (mirror, rep, thr, i)
                    if mirror:
                        hc = h_cnt.numpy()
                        assert hc[0] == cnt[0] and hc[1] == 0
                        assert key(h_out.numpy().view(MATCH_DTYPE)[: hc[0]]) == key(want[(thr, i)])


def test_match_device_back_to_back_frames(oracle, ctx_factory, case1):
    """many frames queued on one stream without host synchronisation in between"""
    import torch

    dev = torch.device("cuda", 0)
    ts = case1["templates"].subset(range(300, 361, 3))
    frame = frame_of(case1)
    ctx = ctx_factory()
    ctx.upload_templates(ts)
    cap = 1024
    stream = torch.cuda.Stream(device=dev)
    d_img = torch.from_numpy(frame).to(dev)
    outs = [torch.zeros(cap * MATCH_DTYPE.itemsize, dtype=torch.uint8, device=dev) for _ in range(4)]
    cnts = [torch.zeros(2, dtype=torch.int32, device=dev) for _ in range(4)]
    with torch.cuda.stream(stream):
        for it in range(40):
            k = it % 4
            ctx.match_device(d_img.data_ptr(), frame.shape[0], frame.shape[1], frame.shape[1] * 3, 3, 88.0,
                             outs[k].data_ptr(), cap, cnts[k].data_ptr(), stream=stream.cuda_stream)
    stream.synchronize()
    pyr = oracle.Pyramid.build(frame, [4, 8], 30.0)
    want = pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, 88.0)
    assert len(want) > 0
    for k in range(4):
        c = cnts[k].cpu().numpy()
        assert c[0] == len(want) and c[1] == 0
        assert key(outs[k].cpu().numpy().view(MATCH_DTYPE)[: c[0]]) == key(want)


def test_match_device_sharded_world1(oracle, ctx_factory, case1):
    """the multi-GPU entry point (template shard + ncclAllGather issued by the library) on a
    one-rank communicator: header + records of the gathered buffer, device and pinned-host copy"""
    import torch

    dev = torch.device("cuda", 0)
    ts = case1["templates"].subset(range(300, 361, 2))
    frame = frame_of(case1)
    ctx = ctx_factory()
    ctx.upload_templates(ts)
    ctx.select_range(5, 20)  # this "rank" owns templates 5..24 of the set
    ctx.comm_init(1, 0, ctx.comm_unique_id())
    cap, hdr, rec = 512, 16, MATCH_DTYPE.itemsize
    stream = torch.cuda.Stream(device=dev)
    d_img = torch.from_numpy(frame).to(dev)
    d_local = torch.zeros(hdr + cap * rec, dtype=torch.uint8, device=dev)
    d_gath = torch.zeros(hdr + cap * rec, dtype=torch.uint8, device=dev)
    h_gath = torch.zeros(hdr + cap * rec, dtype=torch.uint8).pin_memory()
    for _ in range(3):
        with torch.cuda.stream(stream):
            ctx.match_device_sharded(d_img.data_ptr(), frame.shape[0], frame.shape[1], frame.shape[1] * 3, 3, 85.0,
                                     d_local.data_ptr(), cap, d_gath.data_ptr(), gathered_mirror=h_gath.data_ptr(),
                                     stream=stream.cuda_stream)
        stream.synchronize()
    pyr = oracle.Pyramid.build(frame, [4, 8], 30.0)
    shard = ts.subset(range(5, 25))
    want = pyr.match(shard.levels, shard.features, shard.class_idx, shard.template_id, 85.0)
    assert len(want) > 0
    for buf in (d_gath.cpu().numpy(), h_gath.numpy()):
        n, overflow = buf[:8].view(np.int32)
        assert overflow == 0 and n == len(want)
        assert key(buf[hdr:].view(MATCH_DTYPE)[:n]) == key(want)

    # the batched exchange: 3 frames per call, one all-gather of the whole shard (header: 3 pairs, padded to 32 bytes)
    frames = [frame, np.ascontiguousarray(frame[:, ::-1]), np.roll(frame, 40, axis=1)]
    wants = []
    for fr in frames:
        p = oracle.Pyramid.build(fr, [4, 8], 30.0)
        wants.append(p.match(shard.levels, shard.features, shard.class_idx, shard.template_id, 85.0))
        p.free()
    B = len(frames)
    hdrb = (8 * B + 15) // 16 * 16
    nbytes = hdrb + B * cap * rec
    d_imgs = torch.from_numpy(np.stack(frames)).to(dev)
    d_local = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    d_gath = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    h_gath = torch.zeros(nbytes, dtype=torch.uint8).pin_memory()
    fs = frame.shape[0] * frame.shape[1] * 3
    for _ in range(3):
        ctx.match_batch_device_sharded(d_imgs.data_ptr(), fs, B, frame.shape[0], frame.shape[1], frame.shape[1] * 3, 3, 85.0,
                                       d_local.data_ptr(), cap, d_gath.data_ptr(), gathered_mirror=h_gath.data_ptr(),
                                       stream=stream.cuda_stream)
        stream.synchronize()
    for buf in (d_gath.cpu().numpy(), h_gath.numpy()):
        cnt = buf[: 8 * B].view(np.int32).reshape(B, 2)
        for f in range(B):
            assert cnt[f, 1] == 0 and cnt[f, 0] == len(wants[f]), f
            got = buf[hdrb + f * cap * rec: hdrb + (f + 1) * cap * rec].view(MATCH_DTYPE)[: cnt[f, 0]]
            assert key(got) == key(wants[f]), f

    # round 4: the communicator's size as RCCL reports it, and the template loop alone + the same exchange step (how the
    # sharded runs of BASELINE configs 3 and 4 end a step) on the pyramid the last call left resident
    assert ctx.comm_count() == 1
    d_local = torch.zeros(hdr + cap * rec, dtype=torch.uint8, device=dev)
    d_gath = torch.zeros(hdr + cap * rec, dtype=torch.uint8, device=dev)
    h_gath = torch.zeros(hdr + cap * rec, dtype=torch.uint8).pin_memory()
    ctx.match_templates_device_sharded(85.0, d_local.data_ptr(), cap, d_gath.data_ptr(), gathered_mirror=h_gath.data_ptr(),
                                       stream=stream.cuda_stream)
    stream.synchronize()
    for buf in (d_gath.cpu().numpy(), h_gath.numpy()):
        n, overflow = buf[:8].view(np.int32)
        assert overflow == 0 and n == len(wants[0])
        assert key(buf[hdr:].view(MATCH_DTYPE)[:n]) == key(wants[0])


@pytest.mark.parametrize("ch", [3, 1])
def test_match_batch_device(oracle, ctx_factory, case1, ch):
    """sbm_match_batch_device: several different frames in one launch of every kernel; each frame's list must be
    the list of that frame alone (oracle), also after the batch size grows / shrinks and next to single-frame
    calls on the same context, and the pinned-host mirror must hold the same lists."""
    import torch

    dev = torch.device("cuda", 0)
    ts = case1["templates"].subset(range(280, 361, 2))
    base = frame_of(case1)
    frames = [base, np.ascontiguousarray(base[:, ::-1]), np.ascontiguousarray(base[::-1]), np.zeros_like(base),
              np.roll(base, 48, axis=1)]
    if ch == 1:
        frames = [np.ascontiguousarray(f[:, :, 1]) for f in frames]
    rows, cols = frames[0].shape[:2]
    thr = 80.0
    want = []
    for fr in frames:
        pyr = oracle.Pyramid.build(fr, [4, 8], 30.0)
        want.append(pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, thr))
        pyr.free()
    assert len(want[0]) > 0 and len(want[3]) == 0
    ctx = ctx_factory()
    ctx.upload_templates(ts)
    cap = 1024
    rec = MATCH_DTYPE.itemsize
    stream = torch.cuda.Stream(device=dev)
    d_imgs = torch.from_numpy(np.stack(frames)).to(dev)
    fs = rows * cols * ch
    nmax = len(frames)
    d_out = torch.zeros(nmax * cap * rec, dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(nmax * 2, dtype=torch.int32, device=dev)
    h_out = torch.zeros(nmax * cap * rec, dtype=torch.uint8).pin_memory()
    h_cnt = torch.zeros(nmax * 2, dtype=torch.int32).pin_memory()

    def check(n, first, mirror):
        cnt = d_cnt.cpu().numpy().reshape(-1, 2)
        out = d_out.cpu().numpy().reshape(nmax, cap * rec)
        for f in range(n):
            assert cnt[f, 1] == 0 and cnt[f, 0] == len(want[first + f]), (n, first, f, cnt[f], len(want[first + f]))
            assert key(out[f].view(MATCH_DTYPE)[: cnt[f, 0]]) == key(want[first + f])
            if mirror:
                hc = h_cnt.numpy().reshape(-1, 2)
                assert hc[f, 0] == cnt[f, 0] and hc[f, 1] == 0
                assert key(h_out.numpy().reshape(nmax, cap * rec)[f].view(MATCH_DTYPE)[: hc[f, 0]]) == key(want[first + f])

    for mirror in (False, True):
        ctx.set_result_mirror(h_out.data_ptr() if mirror else 0, h_cnt.data_ptr() if mirror else 0)
        for n, first in ((2, 0), (5, 0), (1, 2), (3, 2), (5, 0)):
            d_cnt.fill_(-1)
            h_cnt.fill_(-1)
            torch.cuda.synchronize()  # fills on torch's stream, the match on `stream`
            ctx.match_batch_device(d_imgs.data_ptr() + first * fs, fs, n, rows, cols, cols * ch, ch, thr, d_out.data_ptr(), cap,
                                   d_cnt.data_ptr(), stream=stream.cuda_stream)
            stream.synchronize()
            check(n, first, mirror)
            # a single-frame call on the same context in between
            ctx.match_device(d_imgs.data_ptr() + 1 * fs, rows, cols, cols * ch, ch, thr, d_out.data_ptr(), cap, d_cnt.data_ptr(),
                             stream=stream.cuda_stream)
            stream.synchronize()
            c = d_cnt.cpu().numpy()
            assert c[0] == len(want[1]) and key(d_out.cpu().numpy()[: cap * rec].view(MATCH_DTYPE)[: c[0]]) == key(want[1])


@pytest.mark.parametrize("T", [(4,), (8,), (4, 8, 8)])
def test_match_batch_device_other_pyramids(oracle, ctx_factory, case1, T):
    """the batch entry point on 1- and 3-level pyramids (single level: the coarse pass emits the records itself),
    with a mask shared by the frames and a candidate list that overflows in one frame only"""
    import torch

    dev = torch.device("cuda", 0)
    L = len(T)
    ts_all = case1["templates"]
    # templates restricted to the pyramid depth: level l of the fixture's 2-level templates, coarsest repeated for L = 3
    from shape_based_matching_amd.templates import from_pyramids

    pyrs = []
    for t in range(300, 361, 4):
        lv = []
        for l in range(L):
            src = ts_all.levels[t, min(l, 1)]
            f = ts_all.features[src["feature_offset"]: src["feature_offset"] + src["n_features"]]
            scale = 1 if l < 2 else 2
            feats = np.stack([f["x"] // scale, f["y"] // scale, f["label"]], axis=1)
            lv.append({"width": int(src["width"]) // scale, "height": int(src["height"]) // scale, "tl_x": 0, "tl_y": 0,
                       "pyramid_level": l, "features": feats})
        pyrs.append(lv)
    ts = from_pyramids(pyrs, "t")
    base = frame_of(case1)
    frames = [base, np.roll(base, 64, axis=1), np.zeros_like(base)]
    rows, cols = base.shape[:2]
    mask = np.zeros((rows, cols), np.uint8)
    mask[40:600, 60:700] = 255
    thr = 60.0
    ctx = ctx_factory(T=T)
    ctx.upload_templates(ts)
    cap = 4096
    rec = MATCH_DTYPE.itemsize
    stream = torch.cuda.Stream(device=dev)
    d_imgs = torch.from_numpy(np.stack(frames)).to(dev)
    d_mask = torch.from_numpy(mask).to(dev)
    B = len(frames)
    d_out = torch.zeros(B * cap * rec, dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(B * 2, dtype=torch.int32, device=dev)
    fs = rows * cols * 3
    for use_mask in (False, True):
        want = []
        for fr in frames:
            pyr = oracle.Pyramid.build(fr, list(T), 30.0, mask=mask if use_mask else None)
            want.append(pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, thr))
            pyr.free()
        for _ in range(2):
            ctx.match_batch_device(d_imgs.data_ptr(), fs, B, rows, cols, cols * 3, 3, thr, d_out.data_ptr(), cap, d_cnt.data_ptr(),
                                   stream=stream.cuda_stream, d_mask=d_mask.data_ptr() if use_mask else 0)
            stream.synchronize()
            cnt = d_cnt.cpu().numpy().reshape(B, 2)
            out = d_out.cpu().numpy().reshape(B, cap * rec)
            for f in range(B):
                assert cnt[f, 1] == 0 and cnt[f, 0] == len(want[f]), (T, use_mask, f, cnt[f], len(want[f]))
                assert key(out[f].view(MATCH_DTYPE)[: cnt[f, 0]]) == key(want[f])
        assert len(want[0]) > 0 and len(want[2]) == 0


def test_context_state_sequences(oracle, ctx_factory, case1):
    """one context, a seeded random walk over its entry points (single frame, batch, host path, stage reads, caller-made
    orientation maps, geometry / threshold / template-selection changes): every result against the oracle.  Guards the
    per-level bookkeeping (compact vs 8-plane linear memories, batch capacity, cached thresholds and feature offsets)."""
    import zlib

    import torch

    dev = torch.device("cuda", 0)
    ts = case1["templates"].subset(range(296, 361, 4))
    base = frame_of(case1)                                   # 640 x 768
    small = np.ascontiguousarray(base[64:64 + 512, 96:96 + 576])  # 512 x 576
    geos = {"A": base, "B": small, "Ag": np.ascontiguousarray(base[:, :, 1])}
    ctx = ctx_factory()
    ctx.upload_templates(ts)
    cap, rec = 1024, MATCH_DTYPE.itemsize
    stream = torch.cuda.Stream(device=dev)
    d_out = torch.zeros(4 * cap * rec, dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(8, dtype=torch.int32, device=dev)
    rs = np.random.RandomState(2024)
    pyr_cache = {}

    def want(img, thr):
        k = (img.shape, zlib.crc32(np.ascontiguousarray(img).tobytes()), thr)
        if k not in pyr_cache:
            p = oracle.Pyramid.build(img, [4, 8], 30.0)
            pyr_cache[k] = (p, p.match(ts.levels, ts.features, ts.class_idx, ts.template_id, thr))
        return pyr_cache[k]

    n_checked = 0
    for step in range(40):
        op = rs.randint(0, 6)
        g = ("A", "B", "Ag")[rs.randint(0, 3)]
        img = geos[g]
        ch = 1 if img.ndim == 2 else 3
        thr = (88.0, 75.0)[rs.randint(0, 2)]
        rows, cols = img.shape[:2]
        if op == 0:  # host path
            pyr, w = want(img, thr)
            assert key(ctx.match(img, thr)) == key(w), (step, "match", g, thr)
        elif op == 1:  # single frame, device path
            d_img = torch.from_numpy(img).to(dev)
            ctx.match_device(d_img.data_ptr(), rows, cols, cols * ch, ch, thr, d_out.data_ptr(), cap, d_cnt.data_ptr(), stream=stream.cuda_stream)
            stream.synchronize()
            n = int(d_cnt.cpu().numpy()[0])
            pyr, w = want(img, thr)
            assert key(d_out.cpu().numpy()[: cap * rec].view(MATCH_DTYPE)[:n]) == key(w), (step, "device", g, thr)
        elif op == 2:  # batch of 1..4 frames (shifted copies)
            B = int(rs.randint(1, 5))
            frames = [np.roll(img, 16 * b, axis=1) for b in range(B)]
            d_imgs = torch.from_numpy(np.stack(frames)).to(dev)
            ctx.match_batch_device(d_imgs.data_ptr(), rows * cols * ch, B, rows, cols, cols * ch, ch, thr, d_out.data_ptr(), cap,
                                   d_cnt.data_ptr(), stream=stream.cuda_stream)
            stream.synchronize()
            cnt = d_cnt.cpu().numpy().reshape(4, 2)
            out = d_out.cpu().numpy().reshape(4, cap * rec)
            for b in range(B):
                pyr, w = want(frames[b], thr)
                assert key(out[b].view(MATCH_DTYPE)[: cnt[b, 0]]) == key(w), (step, "batch", g, thr, b)
        elif op == 3:  # read the pyramid state of the last frame-0 build back (expands compact levels)
            ctx.build_pyramid(img)
            pyr = oracle.Pyramid.build(img, [4, 8], 30.0)
            for l in range(2):
                assert np.array_equal(ctx.get_quantized(l), pyr.quantized(l))
                assert np.array_equal(ctx.get_linear_memories(l), pyr.lm(l)), (step, "lm", g, l)
            t = int(rs.randint(0, ts.n_templates))
            for l in range(2):
                cx, cy = int(rs.randint(40, cols - 40)) >> l, int(rs.randint(40, rows - 40)) >> l
                assert np.array_equal(ctx.similarity_local(l, t, cx, cy), pyr.similarity_local(ts.levels[t, l], ts.features, l, cx, cy))
            assert key(ctx.match_templates(thr)) == key(pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, thr))
            pyr.free()
        elif op == 4:  # caller-made orientation maps, template loop only
            pyr, _ = want(img, thr)
            for l in range(2):
                ctx.set_quantized(l, pyr.quantized(l))
            assert key(ctx.match_templates(thr)) == key(pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, thr)), (step, "set_q", g)
        else:  # template selection: a sub-range, then everything again
            first, count = int(rs.randint(0, ts.n_templates - 3)), 3
            ctx.select_range(first, count)
            sub = ts.subset(range(first, first + count))
            pyr, _ = want(img, thr)
            w = pyr.match(sub.levels, sub.features, sub.class_idx, sub.template_id, thr)
            assert key(ctx.match(img, thr)) == key(w), (step, "range", g, first)
            ctx.select_range(0, ts.n_templates)
        n_checked += 1
    assert n_checked == 40
    for p, _ in pyr_cache.values():
        p.free()


def test_batched_stream_config5_geometry(oracle, ctx_factory, case1):
    """BASELINE config 5 geometry (1920 x 1080 cropped to 1920 x 1072, a stream of frames): a batch of 8 frames with the
    object at different places must give, frame by frame, the single-frame entry point's list (all 8) and the oracle's (2)."""
    import torch

    dev = torch.device("cuda", 0)
    ts = case1["templates"].subset(range(0, 361, 3))
    rows, cols = 1072, 1920
    img = case1["test"]
    offs = [(100, 200), (500, 1200), (300, 40), (590, 1310), (0, 0), (250, 700), (400, 1000), (64, 1280)]
    frames = [synth.embed(img, rows, cols, r, c) for (r, c) in offs]
    B = len(frames)
    cap, rec = 512, MATCH_DTYPE.itemsize
    ctx = ctx_factory()
    ctx.upload_templates(ts)
    stream = torch.cuda.Stream(device=dev)
    d_imgs = torch.from_numpy(np.stack(frames)).to(dev)
    d_out = torch.zeros(B * cap * rec, dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(2 * B, dtype=torch.int32, device=dev)
    fs = rows * cols * 3
    ctx.match_batch_device(d_imgs.data_ptr(), fs, B, rows, cols, cols * 3, 3, 90.0, d_out.data_ptr(), cap, d_cnt.data_ptr(),
                           stream=stream.cuda_stream)
    stream.synchronize()
    cnt = d_cnt.cpu().numpy().reshape(B, 2).copy()
    out = d_out.cpu().numpy().reshape(B, cap * rec).copy()
    one_out = torch.zeros(cap * rec, dtype=torch.uint8, device=dev)
    one_cnt = torch.zeros(2, dtype=torch.int32, device=dev)
    total = 0
    for b in range(B):
        ctx.match_device(d_imgs.data_ptr() + b * fs, rows, cols, cols * 3, 3, 90.0, one_out.data_ptr(), cap, one_cnt.data_ptr(),
                         stream=stream.cuda_stream)
        stream.synchronize()
        n = int(one_cnt.cpu().numpy()[0])
        assert cnt[b, 1] == 0 and cnt[b, 0] == n
        assert key(out[b].view(MATCH_DTYPE)[:n]) == key(one_out.cpu().numpy().view(MATCH_DTYPE)[:n]), b
        total += n
    assert total > 0
    for b in (1, 4):
        pyr = oracle.Pyramid.build(frames[b], [4, 8], 30.0)
        want = pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, 90.0)
        assert key(out[b].view(MATCH_DTYPE)[: cnt[b, 0]]) == key(want), b
        pyr.free()


def test_match_batch_device_graph_replay(oracle, ctx_factory, case1):
    """BASELINE config 5's "hipGraph-captured match loop": sbm_match_batch_device in graph mode captures the batch's
    launches once per argument tuple and replays them; lists must equal the oracle's per frame, across replays, after
    the frames' pixels change in place, next to plain-stream calls, and with the pinned result mirror"""
    import torch

    dev = torch.device("cuda", 0)
    ts = case1["templates"].subset(range(280, 361, 2))
    base = frame_of(case1)
    frames = [base, np.ascontiguousarray(base[:, ::-1]), np.roll(base, 48, axis=1), np.zeros_like(base)]
    rows, cols = base.shape[:2]
    thr = 80.0
    want = []
    for fr in frames:
        pyr = oracle.Pyramid.build(fr, [4, 8], 30.0)
        want.append(pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, thr))
        pyr.free()
    assert len(want[0]) > 0
    ctx = ctx_factory()
    ctx.upload_templates(ts)
    cap, rec, B = 1024, MATCH_DTYPE.itemsize, len(frames)
    stream = torch.cuda.Stream(device=dev)
    d_imgs = torch.from_numpy(np.stack(frames)).to(dev)
    d_out = torch.zeros(B * cap * rec, dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(B * 2, dtype=torch.int32, device=dev)
    h_out = torch.zeros(B * cap * rec, dtype=torch.uint8).pin_memory()
    h_cnt = torch.zeros(B * 2, dtype=torch.int32).pin_memory()
    fs = rows * cols * 3

    def run_and_check(order, mirror):
        d_cnt.fill_(-1)
        torch.cuda.synchronize()  # the fill and the frame upload run on torch's stream, the match on `stream`
        ctx.match_batch_device(d_imgs.data_ptr(), fs, B, rows, cols, cols * 3, 3, thr, d_out.data_ptr(), cap, d_cnt.data_ptr(),
                               stream=stream.cuda_stream)
        stream.synchronize()
        cnt = d_cnt.cpu().numpy().reshape(-1, 2)
        out = d_out.cpu().numpy().reshape(B, cap * rec)
        for f in range(B):
            w = want[order[f]]
            assert cnt[f, 1] == 0 and cnt[f, 0] == len(w), (order, f)
            assert key(out[f].view(MATCH_DTYPE)[: cnt[f, 0]]) == key(w)
            if mirror:
                hc = h_cnt.numpy().reshape(-1, 2)
                assert hc[f, 0] == cnt[f, 0]
                assert key(h_out.numpy().reshape(B, cap * rec)[f].view(MATCH_DTYPE)[: hc[f, 0]]) == key(w)

    for mirror in (False, True):
        ctx.set_result_mirror(h_out.data_ptr() if mirror else 0, h_cnt.data_ptr() if mirror else 0)
        for graph in (True, False, True):
            ctx.set_graph_mode(graph)
            order = [0, 1, 2, 3]
            d_imgs.copy_(torch.from_numpy(np.stack([frames[i] for i in order])).to(dev))
            for _ in range(3):  # capture, then replays
                run_and_check(order, mirror)
            order = [2, 3, 0, 1]  # same buffers, other pixels: the replayed graph reads the buffer, not a snapshot
            d_imgs.copy_(torch.from_numpy(np.stack([frames[i] for i in order])).to(dev))
            run_and_check(order, mirror)


def test_graph_replay_is_the_default_with_several_calls_in_flight(oracle, ctx_factory, case1):
    """Round 4 (sbm_set_graph_mode auto): a caller that tells the context it keeps several calls in flight
    (sbm_set_pipeline_depth >= 2) gets captured-graph replay of the batch entry point and of the template-loop entry point
    without asking for it -- from the SECOND sighting of an argument tuple; with one call in flight, with tuples that never
    repeat, or after the opt-out nothing is captured.  Four contexts side by side (the bench's four slots): same lists."""
    import torch

    dev = torch.device("cuda", 0)
    ts = case1["templates"].subset(range(280, 361, 2))
    base = frame_of(case1)
    frames = np.stack([base, np.ascontiguousarray(base[:, ::-1]), np.roll(base, 48, axis=1)])
    rows, cols = base.shape[:2]
    B, cap, rec, thr = len(frames), 1024, MATCH_DTYPE.itemsize, 85.0
    want = []
    for fr in frames:
        pyr = oracle.Pyramid.build(fr, [4, 8], 30.0)
        want.append(key(pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, thr)))
        pyr.free()
    assert len(want[0]) > 0
    d_imgs = [torch.from_numpy(frames).to(dev) for _ in range(3)]
    fs = rows * cols * 3

    class Slot:
        def __init__(self):
            self.ctx = ctx_factory()
            self.ctx.upload_templates(ts)
            self.stream = torch.cuda.Stream(device=dev)
            self.d_out = torch.zeros(B * cap * rec, dtype=torch.uint8, device=dev)
            self.d_cnt = torch.zeros(B * 2, dtype=torch.int32, device=dev)

        def run(self, img):
            self.ctx.match_batch_device(img.data_ptr(), fs, B, rows, cols, cols * 3, 3, thr, self.d_out.data_ptr(), cap, self.d_cnt.data_ptr(),
                                        stream=self.stream.cuda_stream)

        def check(self):
            self.stream.synchronize()
            cnt = self.d_cnt.cpu().numpy().reshape(-1, 2)
            out = self.d_out.cpu().numpy().reshape(B, cap * rec)
            for f in range(B):
                assert cnt[f, 1] == 0 and key(out[f].view(MATCH_DTYPE)[: cnt[f, 0]]) == want[f], f

    slots = [Slot() for _ in range(4)]
    torch.cuda.synchronize()
    s0 = slots[0]
    for _ in range(3):  # one call in flight (the default hint): plain launches
        s0.run(d_imgs[0])
        s0.check()
    assert s0.ctx.graph_count() == 0
    for sl in slots:
        sl.ctx.set_pipeline_depth(4)
    for rep in range(4):  # four contexts round-robin over two input buffers, nothing synchronised in between
        for i, sl in enumerate(slots):
            sl.run(d_imgs[(rep + i) % 2])
        if rep == 0:
            assert all(sl.ctx.graph_count() == 0 for sl in slots)  # first sighting of every tuple: stream launches
    for sl in slots:
        sl.check()
        assert sl.ctx.graph_count() == 2  # one capture per (context, input buffer)
    s0.ctx.set_graph_mode(False)  # the opt-out drops the captures
    s0.run(d_imgs[0])
    s0.check()
    assert s0.ctx.graph_count() == 0
    s0.ctx.set_graph_mode("auto")
    # the template loop alone (BASELINE configs 3 and 4 step this way) on what the last call left resident
    d_out = torch.zeros(cap * rec, dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(2, dtype=torch.int32, device=dev)
    for rep in range(3):
        d_cnt.fill_(-1)
        torch.cuda.synchronize()
        s0.ctx.match_templates_device(thr, d_out.data_ptr(), cap, d_cnt.data_ptr(), stream=s0.stream.cuda_stream)
        s0.stream.synchronize()
        n = int(d_cnt.cpu().numpy()[0])
        assert key(d_out.cpu().numpy().view(MATCH_DTYPE)[:n]) == want[0], rep
        assert s0.ctx.graph_count() == (0 if rep == 0 else 1)


@pytest.mark.parametrize("with_comm", [False, True])
def test_match_batch_device_banded_equals_whole_level_build(oracle, ctx_factory, case1, with_comm):
    """Round 3, build-sharded step on one GPU: the gradient stage launched band by band (2, 4 and 8 row bands, each widened
    by the halo the next level needs) leaves the orientation maps and linear memories of the whole-level build, and the
    match lists of sbm_match_batch_device; with a one-rank communicator the grouped in-place ncclAllGather of the bands
    and the gather of the lists run as well.  BGR and gray, a mask, a frame whose constant canvas crosses band borders."""
    import torch

    dev = torch.device("cuda", 0)
    ts = case1["templates"].subset(range(300, 361, 3))
    base = synth.embed(case1["test"], 640, 768, 80, 80)
    frames = np.stack([base, np.ascontiguousarray(base[:, ::-1]), np.roll(base, 40, axis=1), synth.scene_with_object(5, 640, 768, case1["test"][:473, :600])])
    B, rows, cols = frames.shape[:3]
    cap, rec = 512, MATCH_DTYPE.itemsize
    hdr = (8 * B + 15) // 16 * 16
    nbytes = hdr + B * cap * rec
    stream = torch.cuda.Stream(device=dev)
    wants = []
    for fr in frames:
        p = oracle.Pyramid.build(fr, [4, 8], 30.0)
        wants.append((p.match(ts.levels, ts.features, ts.class_idx, ts.template_id, 85.0), p.quantized(0), p.quantized(1), p.lm(1)))
        p.free()
    assert sum(len(w[0]) for w in wants) > 0
    for ch in (3, 1):
        ctx = ctx_factory()
        ctx.upload_templates(ts)
        if with_comm:
            ctx.comm_init(1, 0, ctx.comm_unique_id())
        fr_in = frames if ch == 3 else np.ascontiguousarray(frames[..., 1])
        d_imgs = torch.from_numpy(fr_in).to(dev)
        fs = rows * cols * ch
        for n_bands in (1, 2, 4, 8):
            d_local = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
            d_gath = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
            h_gath = torch.zeros(nbytes, dtype=torch.uint8).pin_memory()
            torch.cuda.synchronize()
            ctx.match_batch_device_banded(d_imgs.data_ptr(), fs, B, rows, cols, cols * ch, ch, 85.0, d_local.data_ptr(), cap,
                                          d_gathered=d_gath.data_ptr() if with_comm else 0, gathered_mirror=h_gath.data_ptr(),
                                          n_bands=n_bands, stream=stream.cuda_stream)
            stream.synchronize()
            if ch == 3:  # frame 0 of the batch is what the stage read-backs return
                assert np.array_equal(ctx.get_quantized(0), wants[0][1]) and np.array_equal(ctx.get_quantized(1), wants[0][2]), n_bands
                assert np.array_equal(ctx.get_linear_memories(1), wants[0][3]), n_bands
            bufs = [d_local.cpu().numpy(), h_gath.numpy()] + ([d_gath.cpu().numpy()] if with_comm else [])
            # the reference for gray frames: the same engine's whole-level batch call
            if ch == 1:
                d_ref = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
                ctx.match_batch_device(d_imgs.data_ptr(), fs, B, rows, cols, cols, 1, 85.0, d_ref.data_ptr() + hdr, cap, d_ref.data_ptr(),
                                       stream=stream.cuda_stream)
                stream.synchronize()
                ref = d_ref.cpu().numpy()
            for buf in bufs:
                cnt = buf[: 8 * B].view(np.int32).reshape(B, 2)
                for f in range(B):
                    got = buf[hdr + f * cap * rec: hdr + (f + 1) * cap * rec].view(MATCH_DTYPE)[: cnt[f, 0]]
                    if ch == 3:
                        assert cnt[f, 1] == 0 and cnt[f, 0] == len(wants[f][0]), (n_bands, f)
                        assert key(got) == key(wants[f][0]), (n_bands, f)
                    else:
                        rc = ref[: 8 * B].view(np.int32).reshape(B, 2)
                        assert cnt[f].tolist() == rc[f].tolist()
                        assert key(got) == key(ref[hdr + f * cap * rec: hdr + (f + 1) * cap * rec].view(MATCH_DTYPE)[: rc[f, 0]])
        with pytest.raises(capi.SbmError):  # 640 rows do not split into 3 bands
            ctx.match_batch_device_banded(d_imgs.data_ptr(), fs, B, rows, cols, cols * ch, ch, 85.0, d_local.data_ptr(), cap,
                                          d_gathered=d_gath.data_ptr() if with_comm else 0, n_bands=3, stream=stream.cuda_stream)


def test_engine_context_before_torch_cuda_in_a_fresh_process():
    """Round 2 finding: `RuntimeError: No HIP GPUs are available` from torch.cuda.Stream when an engine context was created
    before torch initialised its (bundled, second) HIP runtime.  capi.lib() now maps torch's runtime first, so the process
    has ONE HIP runtime whatever the order -- checked here in the failing order, in a fresh interpreter."""
    import subprocess
    import sys

    from conftest import ROOT

    code = f"""
import sys
sys.path.insert(0, {ROOT!r})
from shape_based_matching_amd import capi
ctx = capi.Context(T=(4, 8), weak_threshold=30.0, device_id=0)
assert "torch" not in sys.modules
import torch
s = torch.cuda.Stream()
x = torch.ones(1 << 16, device="cuda")
torch.cuda.synchronize()
assert float(x.sum()) == float(1 << 16)
files = {{l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l}}
assert len(files) == 1, files
ctx.close()
print("one runtime:", files)
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.parametrize("thr", [0.0, -1.0, 10.0])
def test_strip_plane_refinement_with_clamped_patches(oracle, ctx_factory, case1, thr):
    """the strip-interleaved refinement path (level-0 grid width a multiple of 16, batched entry point)
    with templates that do not fit the frame -- max_x / max_y below the border or negative, so every patch origin is
    clamped (line2Dup.cpp:1227-1245) -- and with thresholds <= 0, where every coarse position is a candidate, the four
    frame corners included."""
    import torch

    dev = torch.device("cuda", 0)
    ts = case1["templates"].subset([0, 90, 340])  # 264 x 266 boxes at level 0
    img = case1["test"]
    frames = [np.ascontiguousarray(img[60:60 + 288, 40:40 + 512]),    # max_y = 288 - 264 - 32 < 0
              np.ascontiguousarray(img[100:100 + 320, 0:320])]        # other frame below; both dimensions too small here
    for fr in frames:
        rows, cols = fr.shape[:2]
        assert cols % 64 == 0 and rows % 16 == 0
        B = 2
        batch = np.stack([fr, np.ascontiguousarray(fr[::-1])])
        ctx = ctx_factory(max_candidates=1 << 16)
        ctx.upload_templates(ts)
        cap, rec = 1 << 15, MATCH_DTYPE.itemsize
        d_imgs = torch.from_numpy(batch).to(dev)
        d_out = torch.zeros(B * cap * rec, dtype=torch.uint8, device=dev)
        d_cnt = torch.zeros(B * 2, dtype=torch.int32, device=dev)
        stream = torch.cuda.Stream(device=dev)
        torch.cuda.synchronize()
        ctx.match_batch_device(d_imgs.data_ptr(), fr.size, B, rows, cols, cols * 3, 3, thr, d_out.data_ptr(), cap, d_cnt.data_ptr(),
                               stream=stream.cuda_stream)
        stream.synchronize()
        cnt = d_cnt.cpu().numpy().reshape(B, 2)
        out = d_out.cpu().numpy().reshape(B, cap * rec)
        for f in range(B):
            p = oracle.Pyramid.build(batch[f], [4, 8], 30.0)
            want = p.match(ts.levels, ts.features, ts.class_idx, ts.template_id, thr)
            p.free()
            assert cnt[f, 1] == 0 and cnt[f, 0] == len(want), (f, cnt[f].tolist(), len(want))
            assert key(out[f].view(MATCH_DTYPE)[: cnt[f, 0]]) == key(want)
        assert len(want) >= 90  # thresholds <= 0 and 10: every coarse position of the (small) span is a candidate


def test_pipeline_depth_hint_changes_launch_sizes_not_results(oracle, ctx_factory, case1):
    """sbm_set_pipeline_depth(3): the gradient launches take fewer, longer work items (32 rows) and the refinement pass 128
    candidate slots per frame; maps, linear memories and match lists are those of the default sizing.  Frame heights that
    are and are not multiples of 32 (the last row block is moved up), a 16-frame batch as in bench.py."""
    import torch

    dev = torch.device("cuda", 0)
    ts = case1["templates"].subset(range(300, 361, 3))
    stream = torch.cuda.Stream(device=dev)
    cap, rec = 512, MATCH_DTYPE.itemsize
    for rows, cols, B in ((640, 768, 16), (656, 1024, 5)):
        base = synth.embed(case1["test"], rows, cols, 80, 80)
        frames = np.stack([np.roll(base, 8 * b, axis=1) for b in range(B)])
        frames[B // 2] = synth.scene_with_object(9, rows, cols, case1["test"])
        d_imgs = torch.from_numpy(frames).to(dev)
        results = []
        for depth in (1, 3):
            ctx = ctx_factory()
            ctx.upload_templates(ts)
            ctx.set_quantize_mode("stream")
            ctx.set_pipeline_depth(depth)
            d_out = torch.zeros(B * cap * rec, dtype=torch.uint8, device=dev)
            d_cnt = torch.zeros(B * 2, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            ctx.match_batch_device(d_imgs.data_ptr(), rows * cols * 3, B, rows, cols, cols * 3, 3, 85.0, d_out.data_ptr(), cap, d_cnt.data_ptr(),
                                   stream=stream.cuda_stream)
            stream.synchronize()
            cnt = d_cnt.cpu().numpy().reshape(B, 2)
            out = d_out.cpu().numpy().reshape(B, cap * rec)
            results.append(([key(out[f].view(MATCH_DTYPE)[: cnt[f, 0]]) for f in range(B)], ctx.get_quantized(0), ctx.get_quantized(1),
                            ctx.get_linear_memories(1)))
        p = oracle.Pyramid.build(frames[0], [4, 8], 30.0)
        assert np.array_equal(results[1][1], p.quantized(0)) and np.array_equal(results[1][2], p.quantized(1))
        assert np.array_equal(results[1][3], p.lm(1))
        p.free()
        assert results[0][0] == results[1][0] and sum(len(x) for x in results[1][0]) > 0
        for f in (B // 2, B - 1):
            p = oracle.Pyramid.build(frames[f], [4, 8], 30.0)
            assert results[1][0][f] == key(p.match(ts.levels, ts.features, ts.class_idx, ts.template_id, 85.0)), f
            p.free()


@pytest.mark.parametrize("T", [(4, 8), (4, 8, 8)])
def test_refinement_as_one_frame_major_list(oracle, ctx_factory, case1, T):
    """sbm_set_refine_order: the refinement pass walking the candidates of up to 64 frames as ONE frame-major list
    (k_similarity_local ORDER 2; the default for batches whose planes exceed the L2s, e.g. BASELINE config 5) against the
    per-frame slots: 70 frames = a group of 64 and a group of 6, frames without any candidate at the start, in the
    middle and at the end of a group, 2- and 3-level pyramids (the 3-level one rewrites the list between the passes),
    and a candidate list that overflows in some frames only (the overflow flag is published per frame)."""
    import torch

    from shape_based_matching_amd.templates import from_pyramids

    dev = torch.device("cuda", 0)
    L = len(T)
    ts_all = case1["templates"]
    pyrs = []
    for t in range(300, 361, 6):
        lv = []
        for l in range(L):
            src = ts_all.levels[t, min(l, 1)]
            f = ts_all.features[src["feature_offset"]: src["feature_offset"] + src["n_features"]]
            scale = 1 if l < 2 else 2
            feats = np.stack([f["x"] // scale, f["y"] // scale, f["label"]], axis=1)
            lv.append({"width": int(src["width"]) // scale, "height": int(src["height"]) // scale, "tl_x": 0, "tl_y": 0,
                       "pyramid_level": l, "features": feats})
        pyrs.append(lv)
    ts = from_pyramids(pyrs, "t")
    rows, cols, B = 512, 640, 70
    base = np.ascontiguousarray(synth.embed(case1["test"], rows, cols, 20, 40)[:, :, 1])
    frames = np.stack([np.roll(base, 4 * (b % 9), axis=1) for b in range(B)])
    for b in (0, 31, 63, 64, 69):
        frames[b] = 0
    d_imgs = torch.from_numpy(frames).to(dev)
    stream = torch.cuda.Stream(device=dev)
    cap, rec, thr = 2048, MATCH_DTYPE.itemsize, 70.0
    got = {}
    for order in ("slots", "list"):
        ctx = ctx_factory(T=T)
        ctx.upload_templates(ts)
        ctx.set_refine_order(order)
        d_out = torch.zeros(B * cap * rec, dtype=torch.uint8, device=dev)
        d_cnt = torch.full((B * 2,), -1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        for _ in range(2):  # the second call finds the counters as the first one left them
            ctx.match_batch_device(d_imgs.data_ptr(), rows * cols, B, rows, cols, cols, 1, thr, d_out.data_ptr(), cap, d_cnt.data_ptr(),
                                   stream=stream.cuda_stream)
            stream.synchronize()
        cnt = d_cnt.cpu().numpy().reshape(B, 2)
        out = d_out.cpu().numpy().reshape(B, cap * rec)
        assert (cnt[:, 1] == 0).all()
        got[order] = [key(out[f].view(MATCH_DTYPE)[: cnt[f, 0]]) for f in range(B)]
    assert got["list"] == got["slots"]
    assert all(len(got["list"][b]) == 0 for b in (0, 31, 63, 64, 69)) and sum(len(x) for x in got["list"]) > B
    for f in (1, 62, 68):
        p = oracle.Pyramid.build(frames[f], list(T), 30.0)
        assert got["list"][f] == key(p.match(ts.levels, ts.features, ts.class_idx, ts.template_id, thr)), f
        p.free()
    # overflow: four candidates per frame kept, flag 1 exactly where the coarse pass found more
    flags = {}
    for order in ("slots", "list"):
        ctx = ctx_factory(T=T, max_candidates=4)
        ctx.upload_templates(ts)
        ctx.set_refine_order(order)
        d_out = torch.zeros(B * cap * rec, dtype=torch.uint8, device=dev)
        d_cnt = torch.full((B * 2,), -1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        ctx.match_batch_device(d_imgs.data_ptr(), rows * cols, B, rows, cols, cols, 1, thr, d_out.data_ptr(), cap, d_cnt.data_ptr(),
                               stream=stream.cuda_stream)
        stream.synchronize()
        cnt = d_cnt.cpu().numpy().reshape(B, 2)
        assert (cnt[:, 0] <= 4).all()
        flags[order] = cnt[:, 1].tolist()
    assert flags["list"] == flags["slots"] and 0 < sum(flags["list"]) < B
