"""-m gpu: the row-streaming gradient kernel (k_quantize_stream) through the C ABI against the oracle, bit for bit
(quantizedOrientations + hysteresisGradient line2Dup.cpp:313-404, :218-311; pyrDown :424-444, checked through the
level-1 orientation map it feeds)."""
import numpy as np
import pytest

from shape_based_matching_amd import synth
from shape_based_matching_amd.templates import MATCH_DTYPE

pytestmark = pytest.mark.gpu


def key(recs):
    return sorted(np.ascontiguousarray(recs, MATCH_DTYPE).tolist())


@pytest.mark.parametrize("ch", [3, 1])
@pytest.mark.parametrize("rows_per_wave", [0, 8, 32, 64])
def test_stream_kernel_levels_match_oracle(oracle, ctx_factory, case1, ch, rows_per_wave):
    """both pyramid levels (level 1 = the kernel's fused pyrDown fed back into it) on a frame with constant
    background, textured regions and strips that end inside the image"""
    rs = np.random.RandomState(7)
    frame = synth.embed(case1["test"], 640, 1008, 80, 200)
    frame[400:640, 0:300] = rs.randint(0, 256, (240, 300, 3))
    if ch == 1:
        frame = np.ascontiguousarray(frame[:, :, 1])
    ctx = ctx_factory()
    ctx.set_quantize_mode("stream", rows_per_wave)
    ctx.upload_templates(case1["templates"].subset(range(300, 361, 6)))
    got = ctx.match(frame, 85.0)
    pyr = oracle.Pyramid.build(frame, [4, 8], 30.0)
    for l in range(2):
        assert np.array_equal(ctx.get_quantized(l), pyr.quantized(l)), l
    ts = case1["templates"].subset(range(300, 361, 6))
    want = pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, 85.0)
    assert key(got) == key(want)
    pyr.free()


@pytest.mark.parametrize("shape", [(16, 16), (32, 48), (48, 240), (64, 244), (80, 484), (96, 724), (112, 1200)])
def test_stream_kernel_awkward_geometries(oracle, ctx_factory, shape):
    rs = np.random.RandomState(shape[1])
    for ch in (1, 3):
        img = rs.randint(0, 256, shape + ((3,) if ch == 3 else ())).astype(np.uint8)
        img[: shape[0] // 3] = 17  # a constant band
        ctx = ctx_factory(T=(4,))
        ctx.set_quantize_mode("stream", 8)
        ctx.build_pyramid(img)
        _, ang, _ = oracle.quantized_orientations(img, 30.0)
        assert np.array_equal(ctx.get_quantized(0), ang), (shape, ch)


def test_stream_kernel_mask_and_batch(oracle, ctx_factory, case1):
    """the batched entry point picks the streaming kernel by itself (mode auto) once the launch is large enough"""
    import torch

    dev = torch.device("cuda", 0)
    ts = case1["templates"].subset(range(280, 361, 4))
    base = synth.embed(case1["test"], 640, 768, 80, 80)
    frames = [np.roll(base, 16 * b, axis=1) for b in range(28)]  # 4 strips x 20 row blocks x 28 frames >= 2048 waves
    mask = np.zeros((640, 768), np.uint8)
    mask[40:600, 60:700] = 255
    ctx = ctx_factory()
    ctx.upload_templates(ts)
    cap, rec, B = 1024, MATCH_DTYPE.itemsize, len(frames)
    stream = torch.cuda.Stream(device=dev)
    d_imgs = torch.from_numpy(np.stack(frames)).to(dev)
    d_mask = torch.from_numpy(mask).to(dev)
    d_out = torch.zeros(B * cap * rec, dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(B * 2, dtype=torch.int32, device=dev)
    for use_mask in (False, True):
        for mode in ("auto", "tile", "stream"):
            ctx.set_quantize_mode(mode)
            d_cnt.fill_(-1)
            torch.cuda.synchronize()  # fill on torch's stream, the match on `stream`
            ctx.match_batch_device(d_imgs.data_ptr(), base.size, B, 640, 768, 768 * 3, 3, 80.0, d_out.data_ptr(), cap, d_cnt.data_ptr(),
                                   d_mask=d_mask.data_ptr() if use_mask else 0, stream=stream.cuda_stream)
            stream.synchronize()
            cnt = d_cnt.cpu().numpy().reshape(-1, 2)
            out = d_out.cpu().numpy().reshape(B, cap * rec)
            for f in (0, 13, 27):
                p = oracle.Pyramid.build(frames[f], [4, 8], 30.0, mask=mask if use_mask else None)
                want = p.match(ts.levels, ts.features, ts.class_idx, ts.template_id, 80.0)
                p.free()
                assert cnt[f, 1] == 0 and cnt[f, 0] == len(want), (use_mask, mode, f)
                assert key(out[f].view(MATCH_DTYPE)[: cnt[f, 0]]) == key(want), (use_mask, mode, f)


def test_stream_kernel_packed_strip_with_frame_stride(oracle, ctx_factory, case1):
    """the narrow last strip of several frames shares a wave (768 columns: 3 x 240 + 48 -> 4 frames per wave), also when
    the caller's frames are not contiguous (every other frame of a buffer: frame stride = 2 frames) and when the batch
    is not a multiple of the group size; every frame's orientation map and match list against the oracle"""
    import torch

    dev = torch.device("cuda", 0)
    ts = case1["templates"].subset(range(300, 361, 6))
    base = synth.embed(case1["test"], 640, 768, 80, 120)  # the object reaches into the last strip
    B = 7
    frames = [np.roll(base, 40 * b, axis=1) for b in range(B)]
    frames[2] = np.full_like(base, 50)  # a constant frame inside a group of textured ones
    buf = np.zeros((2 * B,) + base.shape, np.uint8)
    buf[0::2] = np.stack(frames)
    buf[1::2] = 255 - np.stack(frames)  # the frames in between must not be read
    ctx = ctx_factory()
    ctx.upload_templates(ts)
    ctx.set_quantize_mode("stream", 16)
    cap, rec = 1024, MATCH_DTYPE.itemsize
    stream = torch.cuda.Stream(device=dev)
    d_imgs = torch.from_numpy(buf).to(dev)
    d_out = torch.zeros(B * cap * rec, dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(B * 2, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.match_batch_device(d_imgs.data_ptr(), 2 * base.size, B, 640, 768, 768 * 3, 3, 80.0, d_out.data_ptr(), cap, d_cnt.data_ptr(),
                           stream=stream.cuda_stream)
    stream.synchronize()
    cnt = d_cnt.cpu().numpy().reshape(-1, 2)
    out = d_out.cpu().numpy().reshape(B, cap * rec)
    total = 0
    for f in range(B):
        p = oracle.Pyramid.build(frames[f], [4, 8], 30.0)
        want = p.match(ts.levels, ts.features, ts.class_idx, ts.template_id, 80.0)
        p.free()
        assert cnt[f, 1] == 0 and cnt[f, 0] == len(want), f
        assert key(out[f].view(MATCH_DTYPE)[: cnt[f, 0]]) == key(want), f
        total += len(want)
    assert total > 0
