"""The C-ABI library loads and exports every symbol include/sbm.h declares.
No compute calls: this runs without a GPU."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "sbm.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sbm_[a-z_0-9]+)\s*\(", txt)))


def test_header_and_binding_agree():
    from shape_based_matching_amd import capi

    assert declared_symbols() == sorted(capi.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol():
    from shape_based_matching_amd import capi

    assert os.path.exists(capi.LIB_PATH), "libsbm_hip.so not built: run __graft_entry__.build()"
    L = ctypes.CDLL(capi.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(L, name), name
    assert capi.lib().sbm_abi_version() == 1


def test_every_entry_point_cites_the_reference():
    txt = open(os.path.join(ROOT, "include", "sbm.h")).read()
    assert txt.count("line2Dup.cpp:") >= 15


def test_no_gpu_means_loud_failure():
    """Without a usable GPU sbm_create must fail (no CPU fallback)."""
    import torch

    from shape_based_matching_amd import capi

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.SbmError) as e:
        capi.Context()
    assert e.value.code == -2


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "shape_based_matching_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("the oracle", "").replace("with the oracle", ""), os.path.join(dirpath, f)


def test_hip_runtime_choice_is_checked_by_soname():
    """capi maps torch's bundled HIP runtime before libsbm_hip.so only when its SONAME is the one libsbm_hip.so was linked
    against, and records the choice"""
    import importlib.util

    from shape_based_matching_amd import capi

    needed = [n for n in capi._elf_dynamic_strings(capi.LIB_PATH, 1) if n.startswith("libamdhip64")]
    assert len(needed) == 1 and needed[0].startswith("libamdhip64.so.")
    capi.lib()
    assert capi.HIP_RUNTIME is not None
    spec = importlib.util.find_spec("torch")
    if spec is not None and capi.HIP_RUNTIME != "system":
        assert capi._elf_dynamic_strings(capi.HIP_RUNTIME, 14) == needed
