"""C++ facade, CPU side: the FileStorage YAML subset reader/writer of the drop-in
line2Dup::Detector round-trips the reference's template fixtures (no GPU needed:
readClasses / writeClasses never touch the device)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import REFERENCE, ROOT
from shape_based_matching_amd.templates import read_class_yaml, write_class_yaml

DEMO = os.path.join(ROOT, "shape_based_matching_amd", "sbm_facade_demo")


def same(a, b):
    return np.array_equal(a.levels, b.levels) and np.array_equal(a.features, b.features) and a.class_ids == b.class_ids


@pytest.mark.skipif(not os.path.exists(DEMO), reason="facade demo not built")
def test_cpp_yaml_roundtrip_of_python_written_file(tmp_path, case1):
    ts = case1["templates"].subset(range(0, 361, 40))
    ts.template_id[:] = np.arange(ts.n_templates)
    src = str(tmp_path / "%s_in.yaml")
    write_class_yaml(ts, src % "test")
    out = str(tmp_path / "%s_out.yaml")
    r = subprocess.run([DEMO, "convert", src, "test", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert same(read_class_yaml(out % "test"), ts)


@pytest.mark.skipif(not (os.path.exists(DEMO) and os.path.exists(REFERENCE)), reason="needs the demo and the reference tree")
@pytest.mark.parametrize("case,name", [(0, "circle"), (1, "test"), (2, "test")])
def test_cpp_reads_the_reference_yaml_files(tmp_path, case, name):
    """The reference's own OpenCV-written YAML (test/case*/..._templ.yaml) parsed by the C++ reader."""
    out = str(tmp_path / "%s_out.yaml")
    r = subprocess.run([DEMO, "convert", f"{REFERENCE}/test/case{case}/%s_templ.yaml", name, out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert same(read_class_yaml(out % name), read_class_yaml(f"{REFERENCE}/test/case{case}/{name}_templ.yaml"))


@pytest.mark.skipif(not os.path.exists(DEMO), reason="facade demo not built")
def test_cpp_yaml_gz_roundtrip(tmp_path, case1):
    """templates_%s.yml.gz is the reference's default format (line2Dup.h:310-312)"""
    ts = case1["templates"].subset(range(0, 361, 60))
    ts.template_id[:] = np.arange(ts.n_templates)
    src = str(tmp_path / "%s_in.yaml.gz")
    write_class_yaml(ts, src % "test")  # the Python writer gzips by extension
    out = str(tmp_path / "%s_out.yml.gz")
    r = subprocess.run([DEMO, "convert", src, "test", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    import gzip

    assert gzip.open(out % "test", "rt").readline().startswith("%YAML")
    assert same(read_class_yaml(out % "test"), ts)
