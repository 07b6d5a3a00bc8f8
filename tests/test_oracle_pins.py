"""Pins of the CPU oracle against material the reference itself holds.

The reference cannot be built here (OpenCV is absent), so the oracle is pinned
through the reference's own fixtures:
  * SIMILARITY_LUT (line2Dup.cpp:635): the oracle's closed form is compared with
    the literal table (digest committed; table re-parsed when the reference
    tree is present).
  * test/case1/test_templ.yaml: re-training template 0 from test/case1/train.png
    (test.cpp:262-313) must reproduce all 131 + 71 features, in order; rotating it
    (addTemplate_rotate) must reproduce templates 1..360.  That exercises
    GaussianBlur / Sobel / phase / hysteresisGradient / pyrDown restatements.
"""
import hashlib
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, REFERENCE


def closed_form_lut():
    lut = np.zeros(256, np.uint8)
    for o in range(8):
        for half in range(2):
            for nib in range(16):
                v = nib << (4 * half)
                if v & (1 << o):
                    r = 4
                elif v & ((1 << ((o + 1) & 7)) | (1 << ((o + 7) & 7))):
                    r = 3
                else:
                    r = 0
                lut[32 * o + 16 * half + nib] = r
    return lut


def test_similarity_lut_digest():
    want = open(os.path.join(GOLDEN, "similarity_lut.sha256")).read().strip()
    assert hashlib.sha256(closed_form_lut().tobytes()).hexdigest() == want


@pytest.mark.skipif(not os.path.exists(REFERENCE), reason="reference tree not present")
def test_similarity_lut_against_reference_source():
    txt = open(os.path.join(REFERENCE, "line2Dup.cpp")).read()
    m = re.search(r"SIMILARITY_LUT\[256\]\s*=\s*\{([^}]*)\}", txt)
    lut3 = int(re.search(r"LUT3\s*=\s*(\d+)\s*;", txt).group(1))
    vals = np.array([lut3 if t.strip() == "LUT3" else int(t) for t in m.group(1).split(",")], np.uint8)
    assert np.array_equal(vals, closed_form_lut())


def test_response_maps_equal_lut(oracle):
    """computeResponseMaps (:687): max(LUT[32o + lsb], LUT[32o + 16 + msb]) for all 256 bytes."""
    lut = closed_form_lut()
    sp = np.arange(256, dtype=np.uint8).reshape(16, 16)
    maps = oracle.response_maps(sp)
    for o in range(8):
        want = np.maximum(lut[32 * o + (sp & 15)], lut[32 * o + 16 + (sp >> 4)])
        assert np.array_equal(maps[o], want)


def _case1_training_input(case1):
    """test.cpp:266-279: ROI (130,110,270,270) of train.png, padded by 100, mask likewise."""
    roi = case1["train"][110:380, 130:400]
    padded = np.zeros((470, 470, 3), np.uint8)
    padded[100:370, 100:370] = roi
    mask = np.zeros((470, 470), np.uint8)
    mask[100:370, 100:370] = 255
    return padded, mask


def _same(levels, feats, ts, t):
    for l in range(ts.n_levels):
        lv, ref = levels[l], ts.levels[t, l]
        for k in ("width", "height", "tl_x", "tl_y", "n_features"):
            assert int(lv[k]) == int(ref[k]), (t, l, k)
        mine = feats[int(lv["feature_offset"]) : int(lv["feature_offset"]) + int(lv["n_features"])]
        rf = ts.feats_of(t, l)
        assert np.array_equal(mine["x"], rf["x"]) and np.array_equal(mine["y"], rf["y"]), (t, l)
        assert np.array_equal(mine["label"], rf["label"]), (t, l)


def test_case1_template0_known_answer(oracle, case1):
    padded, mask = _case1_training_input(case1)
    res = oracle.add_template(padded, mask, 2, 128)  # Detector(128, {4, 8}), test.cpp:263
    assert res is not None
    levels, feats = res
    assert [int(l["n_features"]) for l in levels] == [131, 71]
    _same(levels, feats, case1["templates"], 0)


def test_case1_rotated_templates_known_answer(oracle, case1):
    """test.cpp:310-312: addTemplate_rotate(class, 0, angle, centre of the padded image)."""
    padded, mask = _case1_training_input(case1)
    levels, feats = oracle.add_template(padded, mask, 2, 128)
    ts = case1["templates"]
    assert ts.n_templates == 361
    for t in range(1, ts.n_templates):
        ol, of = oracle.add_template_rotate(levels, feats, float(t), (470 / 2.0, 470 / 2.0))
        _same(ol, of, ts, t)


def test_fast_atan2_axes(oracle):
    assert oracle.fast_atan2_deg(0.0, 0.0) == 0.0
    assert abs(oracle.fast_atan2_deg(0.0, 5.0)) < 1e-3
    assert abs(oracle.fast_atan2_deg(5.0, 0.0) - 90.0) < 1e-3
    assert abs(oracle.fast_atan2_deg(0.0, -5.0) - 180.0) < 1e-3
    assert abs(oracle.fast_atan2_deg(-5.0, 0.0) - 270.0) < 1e-3
    for y, x in ((3, 7), (-2, 9), (100, -3), (-50, -50)):
        want = np.degrees(np.arctan2(y, x)) % 360.0
        assert abs(oracle.fast_atan2_deg(float(y), float(x)) - want) < 0.35  # fastAtan2's documented accuracy


@pytest.mark.skipif(not os.path.exists(REFERENCE), reason="reference tree not present")
def test_golden_templates_match_reference_yaml():
    from shape_based_matching_amd.templates import TemplateSet, read_class_yaml

    for case, name in ((0, "circle"), (1, "test"), (2, "test")):
        ts = read_class_yaml(f"{REFERENCE}/test/case{case}/{name}_templ.yaml")
        g = TemplateSet.load_npz(os.path.join(GOLDEN, f"case{case}_templates.npz"))
        assert np.array_equal(ts.levels, g.levels) and np.array_equal(ts.features, g.features)
        assert ts.class_ids == g.class_ids


def test_integer_orientation_rule_equals_float_pipeline(oracle):
    """CPU-side statement of the rule k_quantize uses (tools/derive_orientation_thresholds.py)"""
    g = np.arange(-1020, 1021, dtype=np.int16)
    gx, gy = np.meshgrid(g, g)
    gx, gy = gx.ravel().astype(np.int64), gy.ravel().astype(np.int64)
    ax, ay = np.abs(gx), np.abs(gy)
    mx, mn = np.maximum(ax, ay), np.minimum(ax, ay)
    k = ((mn * 367 >= 73 * mx) & (mx > 0)).astype(np.int64) + ((mn * 395 >= 264 * mx) & (mx > 0))
    k = np.where(ay > ax, 4 - k, k)
    k = np.where(gx < 0, 8 - k, k)
    k = np.where(gy < 0, 16 - k, k)
    assert np.array_equal(k, oracle.orientation_bins(gx.astype(np.int16), gy.astype(np.int16)).astype(np.int64))


# ---- cv::resize (shapeInfo_producer::transform, line2Dup.h:379-405) and the case0 fixture --------------------------
def _resize_linear_numpy(src, fx, fy):
    """independent restatement of OpenCV's 8-bit INTER_LINEAR resize (two 11-bit fixed-point passes)"""
    src = np.ascontiguousarray(src)
    sh, sw = src.shape[:2]
    cn = 1 if src.ndim == 2 else src.shape[2]
    s3 = src.reshape(sh, sw, cn).astype(np.int64)
    dw, dh = int(np.rint(sw * float(fx))), int(np.rint(sh * float(fy)))

    def tab(dn, sn, scale):
        idx = np.zeros(dn, np.int64)
        a0 = np.zeros(dn, np.int64)
        a1 = np.zeros(dn, np.int64)
        for d in range(dn):
            f = np.float32((d + 0.5) * scale - 0.5)
            s = int(np.floor(f))
            f = np.float32(f - np.float32(s))
            if s < 0:
                f, s = np.float32(0), 0
            if s >= sn - 1:
                f, s = np.float32(0), sn - 1
            idx[d] = s
            a0[d] = int(np.rint(np.float32(np.float32(1.0) - f) * np.float32(2048)))
            a1[d] = int(np.rint(f * np.float32(2048)))
        return idx, a0, a1

    xi, xa0, xa1 = tab(dw, sw, 1.0 / float(fx))
    yi, ya0, ya1 = tab(dh, sh, 1.0 / float(fy))
    H = s3[:, xi, :] * xa0[None, :, None] + s3[:, np.minimum(xi + 1, sw - 1), :] * xa1[None, :, None]
    S0, S1 = H[yi], H[np.minimum(yi + 1, sh - 1)]
    out = (((ya0[:, None, None] * (S0 >> 4)) >> 16) + ((ya1[:, None, None] * (S1 >> 4)) >> 16) + 2) >> 2
    out = np.clip(out, 0, 255).astype(np.uint8)
    return out.reshape(dh, dw) if src.ndim == 2 else out


def test_resize_linear_against_numpy_restatement(oracle):
    rs = np.random.RandomState(11)
    for shape in ((37, 53, 3), (64, 64), (5, 9, 3), (200, 150)):
        img = rs.randint(0, 256, shape).astype(np.uint8)
        assert np.array_equal(oracle.resize_linear(img, 1.0, 1.0), img)  # scale 1 is the identity
        for fx in (0.1, 0.37, 0.5, 0.99999934, 1.7, 2.0):
            a, b = oracle.resize_linear(img, fx, fx), _resize_linear_numpy(img, fx, fx)
            assert a.shape == b.shape and np.array_equal(a, b), (shape, fx)


def test_case0_fixture_was_not_made_by_this_fork():
    """test/case0/circle_templ.yaml cannot be re-derived from case0/templ/circle.png with this fork's code (the way
    test/case1/test_templ.yaml is, above), for two reasons the file itself shows:
    1. geometry: every template sits around the centre of the 800 x 800 training canvas whatever its scale (tl + half the
       box = 400 +- 2) -- upstream's transform() warped the image in place with getRotationMatrix2D/warpAffine; this fork
       removed that (line2Dup.h:396-403) and cv::resize's the whole image instead, which moves the object to (400 * scale);
    2. counts: level 0 holds exactly int(150 * scale) features for every scale -- the signature of upstream's older
       selectScatteredFeatures; the fork's version (line2Dup.cpp:163-212) over- or under-shoots (131 / 71 for 128 / 64 in
       case1) and its addTemplate takes the 4th argument as `sscale`, not as the feature count (line2Dup.h:276-286).
    The file therefore stays an INPUT fixture (template data for the matcher), like case2."""
    from shape_based_matching_amd.templates import TemplateSet

    ts = TemplateSet.load_npz(os.path.join(GOLDEN, "case0_templates.npz"))
    scales = np.load(os.path.join(GOLDEN, "case0_info_scales.npy"))
    assert ts.n_templates == len(scales) == 89
    for t in range(ts.n_templates):
        lv = ts.levels[t, 0]
        assert int(lv["n_features"]) == int(np.float32(150) * scales[t]), t
        assert abs(int(lv["tl_x"]) + int(lv["width"]) / 2 - 400) <= 2 and abs(int(lv["tl_y"]) + int(lv["height"]) / 2 - 400) <= 2, t
