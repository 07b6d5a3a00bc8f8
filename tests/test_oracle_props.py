"""The oracle's C code against direct (slow, obviously-correct) numpy/Python
restatements of the reference loops on small inputs, plus structural properties."""
import numpy as np
import pytest

from shape_based_matching_amd import synth
from shape_based_matching_amd.templates import from_pyramids


def brute_spread(q, T):
    rows, cols = q.shape
    out = np.zeros_like(q)
    for r in range(rows):
        for c in range(cols):
            out[r, c] = np.bitwise_or.reduce(q[r : min(r + T, rows), c : min(c + T, cols)].ravel())
    return out


def brute_linearize(m, T):
    rows, cols = m.shape
    out = []
    for rs in range(T):
        for cs in range(T):
            out.append(m[rs::T, cs::T].ravel())
    return np.stack(out)


@pytest.mark.parametrize("T", [2, 4, 5, 8])
def test_spread_linearize_small(oracle, T):
    rs = np.random.RandomState(7 + T)
    rows, cols = 8 * T, 6 * T * 2
    q = synth.onehot_map(rs, rows, cols, 150)
    assert np.array_equal(oracle.spread(q, T), brute_spread(q, T))
    m = rs.randint(0, 5, size=(rows, cols)).astype(np.uint8)
    assert np.array_equal(oracle.linearize(m, T), brute_linearize(m, T))


def test_pyramid_layout_and_tail(oracle):
    rs = np.random.RandomState(3)
    q = synth.onehot_map(rs, 64, 96, 100)
    pyr = oracle.Pyramid.from_quantized([q], [4])
    lm = pyr.lm(0)
    W, H = 96 // 4, 64 // 4
    maps = oracle.response_maps(oracle.spread(q, 4))
    for o in range(8):
        body = lm[o, : 16 * W * H].reshape(16, W * H)
        assert np.array_equal(body, brute_linearize(maps[o], 4))
        assert not lm[o, 16 * W * H :].any()  # zero tail
    assert set(np.unique(lm)) <= {0, 3, 4}


def brute_similarity(lm, lm_stride, rows, cols, T, width, height, feats):
    W, H = cols // T, rows // T
    wf, hf = int((width - 1) / T) + 1, int((height - 1) / T) + 1
    npos = (H - hf) * W + (W - wf) + 1
    dst = np.zeros(W * H, np.int64)
    flat = lm.ravel()
    for (x, y, l) in feats:
        if x < 0 or x >= cols or y < 0 or y >= rows:
            continue
        base = l * lm_stride + ((y % T) * T + (x % T)) * W * H + (y // T) * W + x // T
        if npos > 0:
            dst[:npos] += flat[base : base + npos]
    return dst.reshape(H, W)


def test_similarity_with_row_overrun(oracle):
    """Features at x == width / y == height with width % T == 0 read past the end of a
    linear-memory row into the next one; the flat model must be followed."""
    rs = np.random.RandomState(11)
    rows, cols, T = 192, 256, 8
    q = synth.onehot_map(rs, rows, cols, 120)
    pyr = oracle.Pyramid.from_quantized([q], [T])
    feats = [(64, 48, 1), (0, 0, 2), (64, 0, 3), (0, 48, 4), (13, 27, 5), (300, 5, 6), (63, 47, 0)]
    ts = from_pyramids([[{"width": 64, "height": 48, "features": feats}]])
    got = pyr.similarity(ts.levels[0, 0], ts.features, 0)
    want = brute_similarity(pyr.lm(0), pyr.lm_stride(0), rows, cols, T, 64, 48, feats)
    assert np.array_equal(got.astype(np.int64), want)
    loc = pyr.similarity_local(ts.levels[0, 0], ts.features, 0, 100, 90)
    ox, oy = (100 // T - 8) * T, (90 // T - 8) * T
    W, H = cols // T, rows // T
    flat = pyr.lm(0).ravel()
    wantl = np.zeros((16, 16), np.int64)
    for (x, y, l) in feats:
        x, y = x + ox, y + oy
        if x < 0 or y < 0 or x >= cols or y >= rows:
            continue
        base = l * pyr.lm_stride(0) + ((y % T) * T + (x % T)) * W * H + (y // T) * W + x // T
        for r in range(16):
            wantl[r] += flat[base + r * W : base + r * W + 16]
    assert np.array_equal(loc.astype(np.int64), wantl)


def test_planted_templates_are_found(oracle):
    maps, ts = synth.stage_b(1234, 256, 320, [4, 8], 12, [70, 40], templ_size=96, plant_every=4)
    pyr = oracle.Pyramid.from_quantized(maps, [4, 8])
    recs = pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, 90.0)
    best = {}
    for r in recs:
        best[int(r["template_id"])] = max(best.get(int(r["template_id"]), 0.0), float(r["similarity"]))
    for t in (0, 4, 8):  # planted; two features landing on one pixel cost at most a few points
        assert best.get(t, 0.0) >= 95.0
    c = oracle.canonicalize(recs)
    assert len(c) <= len(recs) and len(oracle.match_set(c)) == len(c)
    sims = c["similarity"]
    assert np.all(sims[:-1] >= sims[1:])
    # threads do not change the multiset (the OpenMP reduction only reorders, :1166-1170)
    recs4 = pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, 90.0, n_threads=4)
    assert sorted(recs.tolist()) == sorted(recs4.tolist())


def test_quantize_structure(oracle):
    img = synth.scene_gray(5, 96, 128, 12)
    mag, ang, ori = oracle.quantized_orientations(img, 30.0)
    assert not ang[0].any() and not ang[-1].any() and not ang[:, 0].any() and not ang[:, -1].any()
    nz = ang[ang != 0]
    assert nz.size > 0 and np.all((nz & (nz - 1)) == 0)  # one-hot
    assert np.all(mag[ang != 0] > 900.0)  # only above weak^2 (:268)
    assert ori.min() >= 0.0 and ori.max() <= 360.0
    # gray image given as 3 equal channels takes the colour branch to the same answer
    mag3, ang3, ori3 = oracle.quantized_orientations(np.stack([img] * 3, axis=2), 30.0)
    assert np.array_equal(ang, ang3) and np.array_equal(mag, mag3) and np.array_equal(ori, ori3)


def test_pyrdown_constant_and_size(oracle):
    img = np.full((37, 50), 93, np.uint8)
    d = oracle.pyrdown(img)
    assert d.shape == (18, 25) and np.all(d == 93)
    rs = np.random.RandomState(2)
    img = rs.randint(0, 256, size=(16, 20)).astype(np.uint8)
    d = oracle.pyrdown(img)
    K = np.array([1, 4, 6, 4, 1])

    def refl(p, n):
        while p < 0 or p >= n:
            p = -p if p < 0 else 2 * n - 2 - p
        return p

    for y in range(8):
        for x in range(10):
            acc = 0
            for j in range(5):
                for i in range(5):
                    acc += K[i] * K[j] * int(img[refl(2 * y + j - 2, 16), refl(2 * x + i - 2, 20)])
            assert d[y, x] == (acc + 128) >> 8
