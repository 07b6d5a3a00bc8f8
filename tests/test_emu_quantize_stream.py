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
    L.sbm_emu_quantize_stream_batch.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_float, vp, vp, C.c_int, C.c_int]

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

    def run_batch(frames, weak=30.0, hs=32, mask=None, pack=1):
        frames = np.ascontiguousarray(frames)
        n, r, c = frames.shape[:3]
        ch = 1 if frames.ndim == 3 else 3
        out = np.full((n, r, c), 0xAA, np.uint8)
        pyr = np.full((n, r // 2, c // 2) + (() if ch == 1 else (3,)), 0xAA, np.uint8)
        lanes = L.sbm_emu_quantize_stream_batch(frames.ctypes.data, n, r, c, c * ch, ch, None if mask is None else mask.ctypes.data,
                                                weak, out.ctypes.data, pyr.ctypes.data, hs, pack)
        assert lanes >= 0
        return out, pyr, lanes

    run.batch = run_batch
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


def check_batch(emu, oracle, frames, expect_lanes, weak=30.0, hs=16, mask=None):
    out, pyr, lanes = emu.batch(frames, weak, hs, mask, pack=1)
    assert lanes == expect_lanes
    for f in range(frames.shape[0]):
        _, ang, _ = oracle.quantized_orientations(frames[f], weak)
        if mask is not None:
            ang = np.where(mask != 0, ang, 0).astype(np.uint8)
        assert np.array_equal(out[f], ang), (f, np.argwhere(out[f] != ang)[:5])
        assert np.array_equal(pyr[f], oracle.pyrdown(frames[f])), f
    # the unpacked launch of the same batch gives the same bytes
    out1, pyr1, lanes1 = emu.batch(frames, weak, hs, mask, pack=0)
    assert lanes1 == 0 and np.array_equal(out1, out) and np.array_equal(pyr1, pyr)


def test_packed_last_strip_of_several_frames(emu, oracle, case1):
    """the last strip of a row (cols - 240 k columns) of up to 64 / (useful lanes + 4) frames shares one wave:
    different frames side by side, each with its own halo lanes, borders, constant runs and pyrDown"""
    rs = np.random.RandomState(11)
    # 272 columns = 240 + 32: last strip 8 useful + 4 halo lanes = 12 -> 5 frames per wave; 7 frames = groups of 5 + 2
    fr = np.stack([synth.scene_bgr(40 + i, 60, 272) for i in range(7)])
    fr[3, :, 200:] = (9, 9, 9)  # a constant right part in one frame only: the wave must not take the shortcut
    fr[5] = 77                  # a completely constant frame next to textured ones
    check_batch(emu, oracle, fr, 12)
    # gray, 304 columns: 64 + 4*... last strip 64 px = 16 + 4 lanes = 20 -> 3 frames per wave; 4 frames
    g = np.stack([rs.randint(0, 256, (40, 304)).astype(np.uint8) for _ in range(4)])
    check_batch(emu, oracle, g, 20, hs=8)
    # one strip only (the strip is first and last: both image borders inside the segments), odd row count, mask
    small = np.stack([synth.scene_bgr(70 + i, 37, 40) for i in range(6)])
    m = np.zeros((37, 40), np.uint8)
    m[5:30, 3:33] = 255
    check_batch(emu, oracle, small, 14, hs=6, mask=m)
    # all frames constant and equal: the packed wave takes the constant-row shortcut
    flat = np.full((5, 48, 272, 3), 31, np.uint8)
    check_batch(emu, oracle, flat, 12)
    # a last strip too wide to share a wave (200 of 240 columns useful): falls back to a wave per frame
    wide = np.stack([synth.scene_gray(90 + i, 32, 440) for i in range(3)])
    out, pyr, lanes = emu.batch(wide, 30.0, 16, None, pack=1)
    assert lanes == 0
    for f in range(3):
        assert np.array_equal(out[f], oracle.quantized_orientations(wide[f], 30.0)[1])


def test_work_items_that_start_in_a_constant_region(emu, oracle):
    """a work item whose rows so far are all one colour skips the arithmetic from its first row on (not only after 11
    rows): texture arriving 1 .. 14 rows after an item's first loaded row, a colour change inside the constant start,
    constant columns only (not a constant row), and the image's top border inside the constant start"""
    rs = np.random.RandomState(21)
    for ch in (1, 3):
        shape = (96, 272) + ((3,) if ch == 3 else ())
        for n in range(1, 15):
            z = rs.randint(0, 256, shape).astype(np.uint8)
            z[16:16 + n] = 90          # a constant band of n rows starting where the 8-row items of rb = 2, 3 load
            z[40:40 + n] = 200 - n     # and another colour further down
            z[60:64] = 33
            z[64:70] = 34              # colour change inside a constant start
            z[70:, :100] = 5           # constant columns, textured rest: never a constant row
            check(emu, oracle, z, hs=8)
            check(emu, oracle, z, hs=16)
        top = rs.randint(0, 256, shape).astype(np.uint8)
        top[:9] = 120                  # the image starts constant: the clamped rows above row 0 are the same colour
        check(emu, oracle, top, hs=8)
        check(emu, oracle, top, hs=32)


@pytest.mark.parametrize("ch", [1, 3])
def test_whole_groups_of_seven_rows_and_overlapping_last_block(emu, oracle, ch):
    """Round 3: a work item runs whole groups of 7 row iterations (no early exit from a group) and the last row block is
    moved up to end at the image's last row.  Rows per item whose 10 warm-up / drain rows do (4, 18, 32, 46) and do not
    (2, 8, 20, 30) fill the last group; heights that are not a multiple of the rows per item, an odd height, an image
    shorter than one item, constant bands crossing the overlap of the last two blocks."""
    rs = np.random.RandomState(31)
    for rows in (50, 61, 96):
        img = rs.randint(0, 256, (rows, 248) + ((3,) if ch == 3 else ())).astype(np.uint8)
        img[rows - 20:rows - 6] = 77  # a constant band inside the rows the last two blocks both compute
        for hs in (2, 4, 8, 18, 20, 30, 32, 46, 130):
            check(emu, oracle, img, hs=hs)


def test_vote_word_of_every_gradient(emu, oracle):
    """the streaming kernel's vote word (four unsigned dot-product sign tests, sign bits added up, a rotate of 1) for all
    2041 x 2041 integer gradients against the 16-bin rule of hysteresisGradient (line2Dup.cpp:225, :243-257: q & 7)"""
    L = C.CDLL(os.path.join(EMU_DIR, "libsbm_emu.so"))
    L.sbm_emu_vote_labels.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    L.sbm_emu_vote_labels.restype = None
    r = np.arange(-1020, 1021, dtype=np.int16)
    gx = np.ascontiguousarray(np.tile(r, len(r)))
    gy = np.ascontiguousarray(np.repeat(r, len(r)))
    got = np.empty(gx.size, np.uint8)
    L.sbm_emu_vote_labels(gx.ctypes.data, gy.ctypes.data, gx.size, got.ctypes.data)
    want = oracle.orientation_bins(gx, gy) & 7
    assert np.array_equal(got, want), np.argwhere(got != want)[:5]
