"""The -DSBM_USE_OPENCV build of the facade (the build an OpenCV host uses; the reference links
OpenCV 4, /root/reference/CMakeLists.txt:36) type-checks against OpenCV 4's public signatures.

This image has no OpenCV, so that branch of include/line2Dup.h had never been through a compiler.
tests/opencv4_api/ holds DECLARATIONS of the OpenCV 4 API subset the facade uses, written from OpenCV's documentation
(own text, no bodies): `g++ -fsyntax-only -DSBM_USE_OPENCV` against them catches every use of something only the bundled
cv:: subset (include/sbm_cvlite.h) offers -- it found a missing <cstring> include that the bundled header had been
supplying.  It says nothing about linking or behaviour: those need a host with OpenCV."""
import os
import subprocess

import pytest

from conftest import ROOT

FACADE = os.path.join(ROOT, "shape_based_matching_amd", "facade")


@pytest.mark.parametrize("src", ["line2Dup_amd.cpp", "nms_c.cpp", "demo.cpp"])
def test_facade_sources_type_check_in_opencv_mode(src):
    r = subprocess.run(["g++", "-std=c++14", "-Wall", "-fsyntax-only", "-DSBM_USE_OPENCV", "-I", os.path.join(ROOT, "tests", "opencv4_api"),
                        "-I", os.path.join(ROOT, "include"), os.path.join(FACADE, src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "error" not in r.stderr


def test_opencv_mode_does_not_pull_in_the_bundled_subset():
    """with SBM_USE_OPENCV the public header includes <opencv2/...> and not sbm_cvlite.h"""
    r = subprocess.run(["g++", "-std=c++14", "-E", "-DSBM_USE_OPENCV", "-I", os.path.join(ROOT, "tests", "opencv4_api"), "-I",
                        os.path.join(ROOT, "include"), os.path.join(ROOT, "include", "line2Dup.h")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "SBM_TEST_OPENCV4_API_CORE_HPP" not in r.stdout  # macro names are consumed by the preprocessor
    assert "sbm_cvlite.h" not in r.stdout and "opencv4_api/opencv2/core.hpp" in r.stdout
