"""The drop-in claim of the boundary (north_star "drops into test.cpp unchanged"): the reference's own
caller, /root/reference/test.cpp, compiles against include/line2Dup.h and links against the facade library, unchanged.
The file is read by path and piped to the compiler (a quoted #include looks in the including file's directory first, so
compiling it in place would pick the reference's own header); nothing of it is copied.  Skipped where the reference
tree does not exist (the GPU box)."""
import os
import subprocess

import pytest

from conftest import REFERENCE, ROOT

TEST_CPP = os.path.join(REFERENCE, "test.cpp")


@pytest.mark.skipif(not os.path.exists(TEST_CPP), reason="reference tree not present")
def test_reference_test_cpp_compiles_and_links_against_the_facade(tmp_path):
    pkg = os.path.join(ROOT, "shape_based_matching_amd")
    assert os.path.exists(os.path.join(pkg, "libsbm_facade.so")), "facade not built: run __graft_entry__.build()"
    src = open(TEST_CPP, "rb").read()
    r = subprocess.run(["g++", "-std=c++14", "-fsyntax-only", "-x", "c++", "-I", os.path.join(ROOT, "include"), "-"], input=src,
                       capture_output=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    assert b"error" not in r.stderr
    exe = str(tmp_path / "ref_test")
    r = subprocess.run(["g++", "-std=c++14", "-x", "c++", "-I", os.path.join(ROOT, "include"), "-", "-o", exe, "-L", pkg, "-lsbm_facade",
                        "-lsbm_hip", f"-Wl,-rpath,{pkg}", f"-Wl,-rpath-link,{pkg}", "-lstdc++fs"], input=src, capture_output=True,
                       cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    assert os.path.exists(exe)
