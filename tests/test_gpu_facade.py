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
