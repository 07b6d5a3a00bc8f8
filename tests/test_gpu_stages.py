"""-m gpu: every HIP stage kernel, called through the C ABI, against the CPU
oracle on the same seeded inputs.  Bit-exact (integer/byte work; the float
outputs of the gradient stage are compared bit for bit as well)."""
import numpy as np
import pytest

from shape_based_matching_amd import synth
from shape_based_matching_amd.templates import from_pyramids

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("shape", [(96, 128), (470, 470), (37, 53), (128, 1024), (16, 64), (3, 3)])
@pytest.mark.parametrize("ch", [1, 3])
def test_quantized_orientations(oracle, ctx_factory, shape, ch):
    ctx = ctx_factory()
    img = synth.scene_gray(10 + ch, *shape, n_shapes=25) if ch == 1 else synth.scene_bgr(20, *shape, n_shapes=25)
    for weak in (30.0, 5.0):
        mag, ang, ori = oracle.quantized_orientations(img, weak)
        gmag, gang, gori = ctx.quantized_orientations(img, weak)
        assert np.array_equal(gang, ang)
        assert np.array_equal(bits(gmag), bits(mag))
        assert np.array_equal(bits(gori), bits(ori))


def test_quantized_orientations_random_noise(oracle, ctx_factory):
    """dense random gradients: exercises every orientation bin and the vote ties"""
    ctx = ctx_factory()
    rs = np.random.RandomState(99)
    for ch in (1, 3):
        shape = (200, 264) if ch == 1 else (200, 264, 3)
        img = rs.randint(0, 256, size=shape).astype(np.uint8)
        mag, ang, ori = oracle.quantized_orientations(img, 10.0)
        gmag, gang, gori = ctx.quantized_orientations(img, 10.0)
        assert np.array_equal(gang, ang)
        assert np.array_equal(bits(gmag), bits(mag)) and np.array_equal(bits(gori), bits(ori))
        assert len(np.unique(ang)) == 9


def test_quantize_case1_train_image(oracle, ctx_factory, case1):
    ctx = ctx_factory()
    img = case1["train"]
    mag, ang, ori = oracle.quantized_orientations(img, 30.0)
    gmag, gang, gori = ctx.quantized_orientations(img, 30.0)
    assert np.array_equal(gang, ang) and np.array_equal(bits(gmag), bits(mag)) and np.array_equal(bits(gori), bits(ori))


@pytest.mark.parametrize("shape,T", [
    ((32, 22), (1, 1)),     # 4-pixel groups straddle the right edge, odd width at level 1
    ((64, 70), (1, 1)),     # one full tile + a 6-column remainder, 35 columns at level 1
    ((96, 66), (1, 1)),
    ((64, 2), (1, 1)),      # narrower than the pyrDown kernel: the literal reflect loop
    ((2, 64), (1, 1)),
    ((16, 11), (1,)),       # odd width, single level
    ((48, 3), (1,)),
    ((144, 208), (1, 1, 1)),  # three levels: 72 x 104, 36 x 52
    ((128, 192), (4, 8)),
])
@pytest.mark.parametrize("ch", [1, 3])
def test_pyramid_awkward_geometries(oracle, ctx_factory, shape, T, ch):
    """the match() gradient path (integer binning, flat-tile shortcut, pyrDown fused into the tile kernel,
    image-border handling of every phase) on frame sizes that are not multiples of the tile or of 4"""
    rs = np.random.RandomState(shape[0] * 1000 + shape[1] + ch)
    noise = rs.randint(0, 256, size=shape if ch == 1 else shape + (3,)).astype(np.uint8)
    scene = synth.scene_gray(3, *shape, n_shapes=12) if ch == 1 else synth.scene_bgr(4, *shape, n_shapes=12)
    flat = np.full_like(noise, 77)  # every tile constant: the shortcut path, pyrDown included
    half = noise.copy()
    half[:, : shape[1] // 2] = 200  # constant tiles next to textured ones
    ctx = ctx_factory(T=T, weak=10.0)
    for img in (noise, scene, flat, half):
        ctx.build_pyramid(img)
        pyr = oracle.Pyramid.build(img, list(T), 10.0)
        for l in range(len(T)):
            assert np.array_equal(ctx.get_quantized(l), pyr.quantized(l)), (shape, ch, l)
        pyr.free()


@pytest.mark.parametrize("shape", [(64, 96), (37, 51), (480, 640), (2, 2)])
@pytest.mark.parametrize("ch", [1, 3])
def test_pyrdown(oracle, ctx_factory, shape, ch):
    ctx = ctx_factory()
    rs = np.random.RandomState(5)
    img = rs.randint(0, 256, size=shape if ch == 1 else shape + (3,)).astype(np.uint8)
    assert np.array_equal(ctx.pyrdown(img), oracle.pyrdown(img))


@pytest.mark.parametrize("T", [1, 2, 4, 5, 8, 16])
def test_spread_response_linearize(oracle, ctx_factory, T):
    ctx = ctx_factory()
    rs = np.random.RandomState(40 + T)
    rows, cols = 16 * T, 16 * T * 3
    q = synth.onehot_map(rs, rows, cols, 80)
    sp = oracle.spread(q, T)
    assert np.array_equal(ctx.spread(q, T), sp)
    maps = oracle.response_maps(sp)
    assert np.array_equal(ctx.compute_response_maps(sp), maps)
    for o in (0, 7):
        assert np.array_equal(ctx.linearize(maps[o], T), oracle.linearize(maps[o], T))


def test_response_maps_all_byte_values(oracle, ctx_factory):
    ctx = ctx_factory()
    sp = np.arange(256, dtype=np.uint8).reshape(16, 16)
    assert np.array_equal(ctx.compute_response_maps(sp), oracle.response_maps(sp))


@pytest.mark.parametrize("T,rows,cols", [(4, 64, 96), (8, 64, 128), (4, 256, 1040), (8, 512, 520), (2, 32, 48), (5, 80, 160),
                                         (16, 64, 128), (4, 1024, 1024), (8, 512, 512)])
def test_build_linear_memories(oracle, ctx_factory, T, rows, cols):
    """fused spread+response+linearize kernel vs the three oracle functions, whole flat block incl. zero tail"""
    ctx = ctx_factory(T=(T,))
    rs = np.random.RandomState(T * 1000 + rows)
    q = synth.onehot_map(rs, rows, cols, 60)
    q[-1, :] = rs.randint(0, 2, cols) << 3  # activity on the last row/col: the clipped window
    q[:, -1] = rs.randint(0, 2, rows) << 5
    ctx.set_quantized(0, q)
    pyr = oracle.Pyramid.from_quantized([q], [T])
    assert np.array_equal(ctx.get_quantized(0), q)
    got = ctx.get_linear_memories(0)
    assert got.shape == (8, pyr.lm_stride(0))
    assert np.array_equal(got, pyr.lm(0))


def test_similarity_maps_and_patches(oracle, ctx_factory):
    rs = np.random.RandomState(77)
    T = (4, 8)
    maps, ts = synth.stage_b(77, 256, 384, T, 6, [131, 71], templ_size=100, plant_every=2, density_permille=60)
    # the overrun corner: features on x == width, y == height with width % T == 0
    ts.levels[1, 1]["width"] = 48
    ts.levels[1, 1]["height"] = 48
    f = ts.features[int(ts.levels[1, 1]["feature_offset"]) :][:6]
    f["x"][:3] = 48
    f["y"][3:6] = 48
    # features outside the image are skipped but still counted
    ts.features["x"][int(ts.levels[2, 1]["feature_offset"])] = 5000
    # a u8-path template (< 64 features) and an empty one
    ts.levels[3, 1]["n_features"] = 20
    ts.levels[4, 1]["n_features"] = 0
    # a template larger than the frame: template_positions <= 0
    ts.levels[5, 1]["width"] = 4000
    ctx = ctx_factory(T=T)
    ctx.upload_templates(ts)
    for l in range(2):
        ctx.set_quantized(l, maps[l])
    pyr = oracle.Pyramid.from_quantized(maps, T)
    for t in range(ts.n_templates):
        want = pyr.similarity(ts.levels[t, 1], ts.features, 1)
        assert np.array_equal(ctx.similarity(t), want), t
        for (cx, cy) in ((32, 32), (101, 77), (200, 150), (383, 255), (40, 250)):
            for l in range(2):
                want = pyr.similarity_local(ts.levels[t, l], ts.features, l, cx >> l, cy >> l)
                assert np.array_equal(ctx.similarity_local(l, t, cx >> l, cy >> l), want), (t, l, cx, cy)


def test_similarity_max_features(oracle, ctx_factory):
    """8191 features (the int16 path's limit, :811): packed accumulation must not overflow"""
    T = (4,)
    rs = np.random.RandomState(1)
    q = np.full((64, 128), 1 << 3, np.uint8)  # every response for label 3 is 4
    nf = 8191
    feats = np.stack([rs.randint(0, 33, nf), rs.randint(0, 33, nf), np.full(nf, 3)], axis=1)
    ts = from_pyramids([[{"width": 32, "height": 32, "features": feats}]])
    ctx = ctx_factory(T=T)
    ctx.upload_templates(ts)
    ctx.set_quantized(0, q)
    pyr = oracle.Pyramid.from_quantized([q], T)
    want = pyr.similarity(ts.levels[0, 0], ts.features, 0)
    got = ctx.similarity(0)
    assert np.array_equal(got, want)
    assert want.max() == 4 * nf
    assert np.array_equal(ctx.similarity_local(0, 0, 64, 40), pyr.similarity_local(ts.levels[0, 0], ts.features, 0, 64, 40))


def test_invalid_arguments_fail_loudly(ctx_factory):
    from shape_based_matching_amd import capi

    ctx = ctx_factory()
    with pytest.raises(capi.SbmError):  # 102 % 4 != 0 (linearize precondition, :751-752)
        ctx.set_quantized(0, np.zeros((100, 102), np.uint8))
    ts = from_pyramids([[{"width": 10, "height": 10, "features": np.zeros((8192, 3), np.int32)}] * 2])
    with pytest.raises(capi.SbmError):  # feature size too large (:1195)
        ctx.upload_templates(ts)
    with pytest.raises(capi.SbmError):  # no pyramid yet
        ctx.match_templates(90.0)


def test_orientation_bins_exhaustive(oracle, ctx_factory):
    """the match path bins orientations with an integer rule instead of fastAtan2: every Sobel gradient
    pair (|gx|, |gy| <= 1020: 2041^2 of them) against the float pipeline of line2Dup.cpp:225/:327"""
    ctx = ctx_factory()
    g = np.arange(-1020, 1021, dtype=np.int16)
    gx, gy = np.meshgrid(g, g)
    want = oracle.orientation_bins(gx.ravel(), gy.ravel())
    got = ctx.orientation_bins(gx.ravel(), gy.ravel())
    assert np.array_equal(got, want)
    assert set(np.unique(want)) == set(range(17))


def test_resize_linear(oracle, ctx_factory):
    """sbm_resize_linear = cv::resize(INTER_LINEAR) of shapeInfo_producer::transform (line2Dup.h:379-405)"""
    rs = np.random.RandomState(21)
    ctx = ctx_factory()
    for shape in ((37, 53, 3), (64, 64), (5, 9, 3), (300, 150), (270, 270, 3)):
        img = rs.randint(0, 256, shape).astype(np.uint8)
        for fx, fy in ((0.1, 0.1), (0.37, 0.37), (0.5, 0.5), (0.99999934, 0.99999934), (1.0, 1.0), (1.7, 1.7), (2.0, 0.5)):
            want = oracle.resize_linear(img, fx, fy)
            if want.size == 0:  # cvRound(rows * fy) == 0: cv::resize asserts, the engine reports an error
                with pytest.raises(Exception):
                    ctx.resize_linear(img, fx, fy)
                continue
            got = ctx.resize_linear(img, fx, fy)
            assert got.shape == want.shape and np.array_equal(got, want), (shape, fx, fy)


def _extract_scan_sequential(mag, thr, mask=None):
    """ColorGradientPyramid::extractTemplate's scan (line2Dup.cpp:452-511) restated literally: row-major walk with the
    magnitude_valid map; returns the pixels whose score survives and exceeds thr^2 (the orientation test is the caller's)"""
    rows, cols = mag.shape
    lm = None
    if mask is not None:  # erode 3x3, BORDER_REPLICATE
        p = np.pad(mask, 1, mode="edge")
        lm = np.min([p[dr:dr + rows, dc:dc + cols] for dr in range(3) for dc in range(3)], axis=0)
    valid = np.ones((rows, cols), bool)
    out = []
    for r in range(2, rows - 2):
        for c in range(2, cols - 2):
            if lm is not None and not lm[r, c]:
                continue
            score = 0.0
            if valid[r, c]:
                score = mag[r, c]
                win = mag[r - 2:r + 3, c - 2:c + 3]
                if (win > score).any():
                    score = 0.0
                else:
                    valid[r - 2:r + 3, c - 2:c + 3] = False
                    valid[r, c] = True
            if score > thr * thr:
                out.append((c, r))
    return out


def test_extract_local_maxima_kernel_equals_the_sequential_scan(ctx_factory, oracle, case1):
    """Round 3: the training-side scan as a HIP kernel + row-major tie resolution.  Plateaus of equal squared magnitude
    (chains where the second pixel is invalidated by the first and the third survives, blocks, diagonal runs, plateaus cut
    by the mask), random integer fields with many ties, a real gradient magnitude image, images too small to scan."""
    ctx = ctx_factory()
    rs = np.random.RandomState(3)
    cases = []
    a = np.zeros((40, 64), np.float32)
    a[10, 5:40] = 5000.0           # a horizontal plateau: every third pixel survives
    a[20:30, 50] = 7000.0          # vertical
    for k in range(12):            # diagonal
        a[25 + k % 6, 8 + k] = 6000.0
    a[32:36, 20:30] = 9000.0       # a block
    cases.append((a, 60.0, None))
    m = np.full((40, 64), 255, np.uint8)
    m[:, 18:22] = 0                # the mask (eroded by one pixel) cuts the horizontal plateau
    cases.append((a, 60.0, m))
    cases.append((rs.randint(0, 6, (48, 80)).astype(np.float32) * 1000.0, 30.0, None))  # ties everywhere
    cases.append((rs.randint(0, 3, (33, 47)).astype(np.float32) * 4000.0, 60.0, (rs.rand(33, 47) > 0.1).astype(np.uint8) * 255))
    mag, _, _ = oracle.quantized_orientations(case1["train"], 30.0)
    cases.append((np.ascontiguousarray(mag[:200, :260]), 60.0, None))
    cases.append((np.ones((5, 5), np.float32) * 1e4, 10.0, None))   # exactly one scanned pixel
    cases.append((np.ones((4, 9), np.float32) * 1e4, 10.0, None))   # nothing to scan
    for mag, thr, mask in cases:
        got = ctx.extract_local_maxima(mag, thr, mask)
        want = _extract_scan_sequential(mag, thr, mask)
        assert [tuple(p) for p in got.tolist()] == want, (mag.shape, thr)
    assert len(_extract_scan_sequential(cases[0][0], 60.0)) > 10
