"""CPU (not gpu): the SOURCE of the gfx950 row-streaming gradient kernel
(shape_based_matching_amd/csrc/sbm_quantize_stream.h), compiled against the wave-level CPU emulation of
tests/emu/wave_emu.h, against the oracle's quantizedOrientations / hysteresisGradient / pyrDown
(line2Dup.cpp:313-404, :218-311, :424-444), bit for bit.  This checks the kernel's dataflow (lane layout, DPP
neighbour exchange, register windows, constant-row shortcut, border handling) where no GPU exists; the -m gpu
tests (tests/test_gpu_quantize_stream.py) then check the real thing through the C ABI."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from shape_based_matching_amd import synth

EMU_DIR = os.path.join(ROOT, "tests", "emu")


@pytest.fixture(scope="module")
def emu():
    subprocess.check_call(["make", "-s", "-C", EMU_DIR])
    L = C.CDLL(os.path.join(EMU_DIR, "libsbm_emu.so"))
    vp = C.c_void_p
    L.sbm_emu_quantize_stream.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_float, vp, vp, C.c_int]

    def run(img, weak=30.0, hs=32, mask=None):
        img = np.ascontiguousarray(img)
        r, c = img.shape[:2]
        ch = 1 if img.ndim == 2 else 3
        out = np.full((r, c), 0xAA, np.uint8)
        pyr = np.full((r // 2, c // 2) + (() if ch == 1 else (3,)), 0xAA, np.uint8)
        rc = L.sbm_emu_quantize_stream(img.ctypes.data, r, c, c * ch, ch, None if mask is None else mask.ctypes.data, weak,
                                       out.ctypes.data, pyr.ctypes.data, hs)
        assert rc == 0
        return out, pyr

    return run


def check(emu, oracle, img, weak=30.0, hs=32, mask=None):
    out, pyr = emu(img, weak, hs, mask)
    _, ang, _ = oracle.quantized_orientations(img, weak)
    if mask is not None:
        ang = np.where(mask != 0, ang, 0).astype(np.uint8)
    assert np.array_equal(out, ang), np.argwhere(out != ang)[:5]
    assert np.array_equal(pyr, oracle.pyrdown(img)), np.argwhere(pyr != oracle.pyrdown(img))[:5]
    return int((ang != 0).sum())


def test_scenes_and_noise(emu, oracle):
    rs = np.random.RandomState(1)
    assert check(emu, oracle, synth.scene_gray(3, 96, 128)) > 1000
    assert check(emu, oracle, synth.scene_bgr(5, 96, 128)) > 1000
    check(emu, oracle, rs.randint(0, 256, (64, 64)).astype(np.uint8))
    check(emu, oracle, rs.randint(0, 256, (70, 260, 3)).astype(np.uint8), hs=16)  # two strips


def test_reference_image_on_black_canvas(emu, oracle, case1):
    """constant background: the constant-row shortcut and its re-entry into the full path"""
    fr = synth.embed(case1["test"], 640, 768, 80, 80)
    assert check(emu, oracle, fr) > 50000
    assert check(emu, oracle, np.ascontiguousarray(fr[:, :, 1])) > 50000


def test_constant_runs_of_changing_colour(emu, oracle):
    rs = np.random.RandomState(2)
    z = np.zeros((200, 500, 3), np.uint8)
    z[0:40] = (10, 20, 30)
    z[40:80] = (200, 100, 50)
    z[80:95] = rs.randint(0, 256, (15, 500, 3))
    z[95:140] = (7, 7, 7)
    z[140:141] = (9, 9, 9)
    z[141:200] = (7, 7, 7)
    check(emu, oracle, z, hs=32)
    check(emu, oracle, np.ascontiguousarray(z[:, :496, 0]), hs=64)
    h = np.zeros((128, 960, 3), np.uint8)  # constant in some strips only
    h[:, 480:] = rs.randint(0, 256, (128, 480, 3))
    h[:, :480] = (50, 60, 70)
    check(emu, oracle, h)


@pytest.mark.parametrize("shape", [(1, 4), (2, 4), (3, 8), (5, 4), (7, 12), (11, 16), (13, 244), (16, 240), (33, 236), (9, 484), (40, 724)])
@pytest.mark.parametrize("ch", [1, 3])
def test_awkward_geometries(emu, oracle, shape, ch):
    rs = np.random.RandomState(shape[0] * 1000 + shape[1] + ch)
    a = rs.randint(0, 256, shape + ((3,) if ch == 3 else ())).astype(np.uint8)
    check(emu, oracle, a, hs=2 if shape[0] < 8 else 16)


def test_mask_thresholds_and_rows_per_wave(emu, oracle):
    rs = np.random.RandomState(4)
    m = np.zeros((96, 128), np.uint8)
    m[10:80, 20:100] = 255
    m[30:40, 30:50] = 0
    check(emu, oracle, synth.scene_bgr(5, 96, 128), mask=m)
    g = (synth.scene_gray(9, 64, 256) + rs.randint(0, 3, (64, 256))).astype(np.uint8)
    for weak in (0.0, 5.5, 30.0, 200.0):
        check(emu, oracle, g, weak=weak)
    for hs in (2, 6, 10, 64, 128):
        check(emu, oracle, synth.scene_bgr(11, 100, 248), hs=hs)


def test_extreme_gradients(emu, oracle):
    e = np.zeros((64, 64, 3), np.uint8)
    e[:, ::2] = 255
    check(emu, oracle, e)
    e = np.zeros((64, 64), np.uint8)
    e[::2, :] = 255
    check(emu, oracle, e)
    check(emu, oracle, (np.indices((64, 64)).sum(0) % 2 * 255).astype(np.uint8))


@pytest.mark.parametrize("ch", [1, 3])
def test_constant_run_fast_forward_chunks(emu, oracle, ch):
    """long constant runs are consumed seven rows at a time; a textured band starting at every phase of that chunking
    must end the run on the right row"""
    rs = np.random.RandomState(5)
    for start in range(40, 54):
        img = np.full((140, 256) + ((3,) if ch == 3 else ()), 90, np.uint8)
        img[start:start + 9] = rs.randint(0, 256, (9,) + img.shape[1:])
        img[120:] = 200  # a second colour: a new run after one changing row
        for hs in (32, 64, 140):
            check(emu, oracle, img, hs=hs)
