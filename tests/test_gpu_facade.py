"""-m gpu: the drop-in C++ line2Dup::Detector (include/line2Dup.h) driven through the demo binary."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from shape_based_matching_amd import synth
from shape_based_matching_amd.templates import read_class_yaml, write_class_yaml

pytestmark = pytest.mark.gpu
DEMO = os.path.join(ROOT, "shape_based_matching_amd", "sbm_facade_demo")


def write_ppm(path, bgr):
    rgb = np.ascontiguousarray(bgr[:, :, ::-1])
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (rgb.shape[1], rgb.shape[0]))
        f.write(rgb.tobytes())


def write_pgm(path, g):
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (g.shape[1], g.shape[0]))
        f.write(np.ascontiguousarray(g).tobytes())


def test_facade_match_case1(tmp_path, oracle, case1):
    """test.cpp:angle_test("test"): readClasses -> pad 250 -> crop to x16 -> match(img, 90, ids)"""
    assert os.path.exists(DEMO), "facade demo not built: run __graft_entry__.build()"
    ts = case1["templates"]
    fmt = str(tmp_path / "%s_templ.yaml")
    write_class_yaml(ts, fmt % "test")
    img_path = str(tmp_path / "test.ppm")
    write_ppm(img_path, case1["test"])
    r = subprocess.run([DEMO, "match", fmt, "test", img_path, "90", "128", "250"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    assert lines[0].startswith("matches ")
    got = [tuple(l.split()) for l in lines[1:]]
    got = [(int(a), int(b), int(c), d, int(e)) for a, b, c, d, e in got]

    img = case1["test"]
    p = synth.embed(img, img.shape[0] + 500, img.shape[1] + 500, 250, 250)
    frame = np.ascontiguousarray(p[: p.shape[0] // 16 * 16, : p.shape[1] // 16 * 16])
    pyr = oracle.Pyramid.build(frame, [4, 8], 30.0)
    want = oracle.canonicalize(pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, 90.0))
    # the reference's adjacent std::unique: equal (x, y, similarity, class_id) neighbours collapse
    keep = []
    for m in want:
        k = (int(m["x"]), int(m["y"]), int(m["similarity"].view(np.uint32)), "test")
        if keep and keep[-1][:4] == k:
            continue
        keep.append(k + (int(m["template_id"]),))
    assert got == keep
    assert len(got) > 0 and got[0][4] == 340


def test_facade_training_reproduces_reference_fixture(tmp_path, case1):
    """addTemplate + addTemplate_rotate through the product (HIP quantize / pyrDown kernels + host
    feature selection) must reproduce the reference's own test/case1/test_templ.yaml"""
    assert os.path.exists(DEMO)
    roi = case1["train"][110:380, 130:400]  # test.cpp:266-279
    padded = np.zeros((470, 470, 3), np.uint8)
    padded[100:370, 100:370] = roi
    mask = np.zeros((470, 470), np.uint8)
    mask[100:370, 100:370] = 255
    write_ppm(str(tmp_path / "train.ppm"), padded)
    write_pgm(str(tmp_path / "mask.pgm"), mask)
    fmt = str(tmp_path / "%s_out.yaml")
    r = subprocess.run([DEMO, "train", str(tmp_path / "train.ppm"), str(tmp_path / "mask.pgm"), "128", "360", "1", fmt, "test",
                        str(tmp_path / "info.yaml")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = read_class_yaml(fmt % "test")
    ref = case1["templates"]
    assert got.n_templates == 361
    assert np.array_equal(got.levels, ref.levels)
    assert np.array_equal(got.features, ref.features)
    info = open(str(tmp_path / "info.yaml")).read()
    assert info.count("angle") == 361


def _parse_matches(lines):
    got = [tuple(l.split()) for l in lines]
    return [(int(a), int(b), int(c), d, int(e)) for a, b, c, d, e in got]


def test_facade_scale_train(tmp_path, oracle, golden):
    """test.cpp:scale_test("train") through the facade: shapeInfo_producer::src_of / mask_of with scale != 1
    (cv::resize, line2Dup.h:379-405) feeding addTemplate (HIP gradient kernels + host selection), against the
    oracle's resize + addTemplate"""
    from shape_based_matching_amd.templates import TemplateSet

    assert os.path.exists(DEMO)
    img = np.load(os.path.join(golden, "case0_circle_bgr.npz"))["bgr"]
    write_ppm(str(tmp_path / "circle.ppm"), img)
    fmt = str(tmp_path / "%s_templ.yaml")
    r = subprocess.run([DEMO, "scale_train", str(tmp_path / "circle.ppm"), "150", "0.25", "1", "0.25", fmt, "circle",
                        str(tmp_path / "info.yaml")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = read_class_yaml(fmt % "circle")
    # produce_infos: for (scale = lo; scale <= hi + eps; scale += step) in float (line2Dup.h:424-428)
    scales, s = [], np.float32(0.25)
    while s <= np.float32(1) + np.float32(0.00001):
        scales.append(s)
        s = np.float32(s + np.float32(0.25))
    assert len(scales) == 4
    t = 0
    for sc in scales:
        src = oracle.resize_linear(img, float(sc), float(sc))
        mask = np.full(src.shape[:2], 255, np.uint8)
        res = oracle.add_template(src, mask, 2, 150)  # the fork's addTemplate: 4th argument is sscale, the count stays 150
        if res is None:
            continue
        levels, feats = res
        for l in range(2):
            lv, ref = levels[l], got.levels[t, l]
            for k in ("width", "height", "tl_x", "tl_y", "n_features"):
                assert int(lv[k]) == int(ref[k]), (float(sc), l, k)
            mine = feats[int(lv["feature_offset"]): int(lv["feature_offset"]) + int(lv["n_features"])]
            rf = got.feats_of(t, l)
            assert np.array_equal(mine["x"], rf["x"]) and np.array_equal(mine["y"], rf["y"]) and np.array_equal(mine["label"], rf["label"])
        t += 1
    assert t == got.n_templates and t >= 3
    assert open(str(tmp_path / "info.yaml")).read().count("scale") == t


def test_facade_nms_flow_case2(tmp_path, oracle, case2):
    """test.cpp:noise_test (455-491): Detector(30, {4,8}) -> match(img, 90) -> boxes from templ[0] -> NMSBoxes(0, 0.5)"""
    from test_nms import py_nms

    assert os.path.exists(DEMO)
    ts = case2["templates"]
    fmt = str(tmp_path / "%s_templ.yaml")
    write_class_yaml(ts, fmt % "test")
    img = case2["test"]
    write_ppm(str(tmp_path / "test.ppm"), img)
    r = subprocess.run([DEMO, "nms", fmt, "test", str(tmp_path / "test.ppm"), "90", "30", "0"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    ms = [tuple(int(v) for v in l.split()[1:]) for l in lines if l.startswith("m ")]
    kept = [int(l.split()[1]) for l in lines if l.startswith("k ")]
    assert len(ms) > 0 and lines[0] == f"matches {len(ms)} kept {len(kept)}"
    # the match list itself: oracle, canonical order, the reference's adjacent unique
    frame = np.ascontiguousarray(img[: img.shape[0] // 16 * 16, : img.shape[1] // 16 * 16])
    pyr = oracle.Pyramid.build(frame, [4, 8], 30.0)
    want = oracle.canonicalize(pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, 90.0))
    keep = []
    for m in want:
        k = (int(m["x"]), int(m["y"]), int(m["similarity"].view(np.uint32)))
        if keep and keep[-1][:3] == k:
            continue
        keep.append(k + (int(m["template_id"]),))
    assert [m[:4] for m in ms] == keep
    # boxes are (x, y, width, height of level 0 of the matched template), scores the similarities
    for m in ms:
        assert (m[4], m[5]) == (int(ts.levels[m[3], 0]["width"]), int(ts.levels[m[3], 0]["height"]))
    boxes = [[m[0], m[1], m[4], m[5]] for m in ms]
    scores = [float(np.uint32(m[2]).view(np.float32)) for m in ms]
    assert kept == py_nms(boxes, scores, 0.0, 0.5)
    assert 0 < len(kept) < len(ms)


def test_facade_get_instance(tmp_path, case1):
    """Detector::getInstance(path) (line2Dup.cpp:1366-1393): detector settings + `classes` + `templates_dir` from one
    YAML, templates from <dir>/<class>.yaml.gz; the singleton is built once; matches equal the explicit flow's"""
    import gzip

    assert os.path.exists(DEMO)
    ts = case1["templates"].subset(range(300, 361, 2))
    ts.template_id[:] = np.arange(ts.n_templates)  # readClass asserts template_id == index (:1532)
    tdir = tmp_path / "templates"
    tdir.mkdir()
    plain = str(tmp_path / "test_plain.yaml")
    write_class_yaml(ts, plain)
    with open(plain, "rb") as f, gzip.open(str(tdir / "test.yaml.gz"), "wb") as g:
        g.write(f.read())
    cfg = str(tmp_path / "detector_linemod.yaml")
    with open(cfg, "w") as f:
        f.write("%YAML:1.0\n---\npyramid_levels: 2\nT: [ 4, 8 ]\ntype: ColorGradient\nweak_threshold: 30.\nnum_features: 128\n"
                f"strong_threshold: 60.\ntemplates_dir: \"{tdir}\"\nclasses:\n   - \"test\"\n")
    frame = synth.embed(case1["test"], 640, 768, 80, 80)
    write_ppm(str(tmp_path / "frame.ppm"), frame)
    r = subprocess.run([DEMO, "instance", cfg, str(tmp_path / "frame.ppm"), "88"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    head = [l for l in lines if l.startswith("instance ")][0]
    assert head.startswith(f"instance same 1 classes 1 templates {ts.n_templates} T 4 8 matches ")
    got = _parse_matches(lines[lines.index(head) + 1:])
    fmt = str(tmp_path / "%s_plain.yaml")
    r2 = subprocess.run([DEMO, "match", fmt, "test", str(tmp_path / "frame.ppm"), "88", "128", "0"], capture_output=True, text=True)
    assert r2.returncode == 0, r2.stderr
    assert got == _parse_matches(r2.stdout.strip().splitlines()[1:]) and len(got) > 0
    # a missing configuration file throws (line2Dup.cpp:1369-1373)
    r3 = subprocess.run([DEMO, "instance", str(tmp_path / "nope.yaml"), str(tmp_path / "frame.ppm"), "88"], capture_output=True, text=True)
    assert r3.returncode != 0


def test_facade_batch_async_and_devices(tmp_path, oracle, case1):
    """Round 3 extensions of the C++ Detector: matchBatch / matchAsync + wait (sub-batches of 8: 20 frames = 8 + 8 + 4, both
    upload buffers re-used) and setDevices({0, 0, 0}) -- match() shards the templates over three contexts and host threads,
    matchBatch() deals the frames over them -- each against per-frame match() in the demo, frame 0 against the oracle here"""
    ts = case1["templates"].subset(range(280, 361, 2))
    ts.class_ids = ["test"]
    ts.template_id = np.arange(ts.n_templates, dtype=np.int32)
    fmt = str(tmp_path / "%s_templ.yaml")
    write_class_yaml(ts, fmt % "test")
    img_path = str(tmp_path / "test.ppm")
    write_ppm(img_path, case1["test"])
    for nf, devs in ((20, "0,0,0"), (3, "0,0")):
        r = subprocess.run([DEMO, "batch", fmt, "test", img_path, "88", "128", str(nf), "100", devs], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        lines = r.stdout.strip().splitlines()
        head = lines[0].split()
        flags = dict(zip(head[1::2], head[2::2]))
        assert flags["frames"] == str(nf) and int(flags["matches"]) > 0
        for k in ("batch_same", "async_same", "devices_same", "devices_batch_same", "unknown_class_empty", "async_interleaved_same"):
            assert flags[k] == "1", (k, lines[0])
        got = [tuple(l.split()) for l in lines[1:]]
        got = [(int(a), int(b), int(c), d, int(e)) for a, b, c, d, e in got]
        img = case1["test"]
        p = synth.embed(img, img.shape[0] + 200, img.shape[1] + 200, 100, 100)
        frame = np.ascontiguousarray(p[: p.shape[0] // 16 * 16, : p.shape[1] // 16 * 16])
        pyr = oracle.Pyramid.build(frame, [4, 8], 30.0)
        want = oracle.canonicalize(pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, 88.0))
        pyr.free()
        keep = []
        for m in want:
            k = (int(m["x"]), int(m["y"]), int(m["similarity"].view(np.uint32)), "test")
            if keep and keep[-1][:4] == k:
                continue
            keep.append(k + (int(m["template_id"]),))
        assert got == keep and len(got) > 0


def test_facade_concurrent_callers_on_one_detector(tmp_path, case1):
    """Round 4: Detector::match() const is re-entrant, as the reference's (line2Dup.h:272-274, line2Dup.cpp:1078-1150 keep
    no state): 4 host threads x 50 calls on ONE detector, two frame sizes interleaved, a matchBatch every tenth call; every
    list equals the one the call returns alone.  Once with a lane per thread, once with 2 lanes (callers wait), once with 1
    (serialised)."""
    ts = case1["templates"].subset(range(300, 361, 3))
    ts.class_ids = ["test"]
    ts.template_id = np.arange(ts.n_templates, dtype=np.int32)
    fmt = str(tmp_path / "%s_templ.yaml")
    write_class_yaml(ts, fmt % "test")
    img_path = str(tmp_path / "test.ppm")
    write_ppm(img_path, case1["test"])
    for lanes in ("4", "2", "1"):
        r = subprocess.run([DEMO, "threads", fmt, "test", img_path, "88", "128", "4", "50", "60", lanes], capture_output=True, text=True,
                           timeout=600)
        assert r.returncode == 0, (lanes, r.stdout, r.stderr)
        head = r.stdout.strip().splitlines()[-1].split()
        # "threads 4 calls 50 matches <n0> <n1> different 0"
        assert head[:5] == ["threads", "4", "calls", "50", "matches"] and head[7:] == ["different", "0"], r.stdout
        assert int(head[5]) > 0 and int(head[6]) > 0  # both geometries find the object


def test_facade_match_batch_lists_longer_than_the_batch_capacity(tmp_path, oracle, case1):
    """matchBatch() promises lists[f] == match(frames[f]); a frame whose RAW (pre-dedup) record count exceeds
    the batch's per-frame capacity of 1024 must be matched again with a larger buffer, not turned into an exception"""
    ts = case1["templates"]
    fmt = str(tmp_path / "%s_templ.yaml")
    write_class_yaml(ts, fmt % "test")
    img_path = str(tmp_path / "test.ppm")
    write_ppm(img_path, case1["test"])
    img = case1["test"]
    p = synth.embed(img, img.shape[0] + 200, img.shape[1] + 200, 100, 100)
    frame = np.ascontiguousarray(p[: p.shape[0] // 16 * 16, : p.shape[1] // 16 * 16])
    pyr = oracle.Pyramid.build(frame, [4, 8], 30.0)
    thr = None
    for t in (65.0, 60.0):
        raw = pyr.match(ts.levels, ts.features, ts.class_idx, ts.template_id, t, n_threads=min(16, os.cpu_count() or 1))
        if len(raw) > 1100:
            thr = t
            break
    pyr.free()
    assert thr is not None, "no threshold gave more than 1024 raw records"
    r = subprocess.run([DEMO, "batch", fmt, "test", img_path, str(thr), "128", "5", "100", "0,0"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    head = r.stdout.strip().splitlines()[0].split()
    flags = dict(zip(head[1::2], head[2::2]))
    for k in ("batch_same", "async_same", "devices_same", "devices_batch_same"):
        assert flags[k] == "1", (k, head)
