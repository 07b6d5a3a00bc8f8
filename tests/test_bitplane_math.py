"""CPU (not gpu): the register-level arithmetic of the coarse pass on bit planes
(shape_based_matching_amd/csrc/sbm_bitplane_math.h, host pass compiled by tests/emu) against numpy.

The reference adds one response byte per (position, feature) (similarity, line2Dup.cpp:843-856); the bit-plane kernels
replace that by bit counts: the producer turns 32 spread bytes into 8 dwords of orientation bits, the consumer counts
bits with bit-sliced carry-save counters whose top carry is "more misses than the threshold allows"."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

EMU_DIR = os.path.join(ROOT, "tests", "emu")


@pytest.fixture(scope="module")
def emu():
    subprocess.check_call(["make", "-s", "-C", EMU_DIR])
    L = C.CDLL(os.path.join(EMU_DIR, "libsbm_emu.so"))
    L.sbm_emu_bytes32_to_bitplanes.argtypes = [C.c_void_p, C.c_void_p]
    L.sbm_emu_bitslice_count.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    L.sbm_emu_bitslice_count.restype = C.c_uint32
    L.sbm_emu_bitslice_count32.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    L.sbm_emu_bitslice_count32.restype = C.c_uint32
    L.sbm_emu_local_row_best.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.sbm_emu_local_row_best.restype = C.c_uint32
    return L


def test_bytes32_to_bitplanes(emu):
    rs = np.random.RandomState(5)
    cases = [rs.randint(0, 256, 32).astype(np.uint8) for _ in range(500)]
    cases += [np.zeros(32, np.uint8), np.full(32, 255, np.uint8), np.arange(32, dtype=np.uint8), (1 << (np.arange(32) % 8)).astype(np.uint8)]
    for cells in cases:
        out = np.zeros(8, np.uint32)
        emu.sbm_emu_bytes32_to_bitplanes(cells.ctypes.data, out.ctypes.data)
        for o in range(8):
            want = int(sum(((int(cells[b]) >> o) & 1) << b for b in range(32)))
            assert int(out[o]) == want, (o, cells)


@pytest.mark.parametrize("P", [3, 7, 10, 13])
def test_bitsliced_counters_and_sticky_overflow(emu, P):
    rs = np.random.RandomState(P)
    for n, density in ((8, 0.5), (64, 0.3), (72, 0.9), (1024, 0.62), (8192, 0.1), (8192, 0.97)):
        if n > 8 * ((1 << P) - 1) and P < 13:
            n = 8 * ((1 << P) - 1)
        x = np.zeros(n, np.uint32)
        bits = rs.rand(n, 32) < density
        for b in range(32):
            x |= bits[:, b].astype(np.uint32) << np.uint32(b)
        total = bits.sum(axis=0)
        for bias in (0, 1, (1 << P) - 1, rs.randint(0, 1 << P)):
            counts = np.zeros(32, np.int32)
            dead = emu.sbm_emu_bitslice_count(x.ctypes.data, n, P, bias, counts.ctypes.data, 0)
            assert np.array_equal(counts, (bias + total) % (1 << P))
            # the sticky carry = "the counter passed 2^P at some point" = bias + count >= 2^P (counts only grow)
            want_dead = sum(1 << b for b in range(32) if bias + total[b] >= (1 << P))
            assert dead == want_dead, (n, density, bias)
            if P >= 7 and n % 32 == 0:  # the 32-slice form (one ripple per 32 slices) counts the same
                counts32 = np.zeros(32, np.int32)
                dead32 = emu.sbm_emu_bitslice_count32(x.ctypes.data, n, P, bias, counts32.ctypes.data, 0)
                assert np.array_equal(counts32, counts) and dead32 == want_dead, (n, density, bias)
            # inverted-input form: the zero bits of the complemented slices are the same set
            xi = ~x
            ci = np.zeros(32, np.int32)
            di = emu.sbm_emu_bitslice_count(xi.ctypes.data, n, P, bias, ci.ctypes.data, 1)
            assert np.array_equal(ci, counts) and di == want_dead, (n, density, bias)


@pytest.mark.parametrize("P,nfs", [(6, [1, 2, 3, 7, 8, 9, 17, 33, 63, 64, 65, 71, 100, 131, 252]), (9, [253, 257, 600, 1021, 2044]),
                                   (12, [2045, 4095, 8191])])
def test_refinement_row_on_bit_strips(emu, P, nfs):
    """one patch row of the refinement pass on bit strips (sbm_local_bits.h): per feature 16 "response > 0" bits and 16
    "response == 4" bits; the reference's sum of response bytes {0, 3, 4} (similarityLocal, line2Dup.cpp:860-922) is
    3 #any + #exact, its first maximum the lowest column with the largest sum"""
    rs = np.random.RandomState(P)
    for nf in nfs:
        for d_any, d_ex in ((0.5, 0.5), (0.95, 0.9), (0.05, 0.5), (1.0, 1.0), (0.0, 0.0)):
            any_b = rs.rand(nf, 16) < d_any
            ex_b = any_b & (rs.rand(nf, 16) < d_ex)  # an exact hit is also a hit
            x = np.zeros(nf, np.uint32)
            for col in range(16):
                x |= any_b[:, col].astype(np.uint32) << np.uint32(col)
                x |= ex_b[:, col].astype(np.uint32) << np.uint32(16 + col)
            raw = 3 * any_b.sum(axis=0) + ex_b.sum(axis=0)
            first = C.c_int(-1)
            got = emu.sbm_emu_local_row_best(x.ctypes.data, nf, P, C.byref(first))
            assert got == raw.max(), (nf, d_any)
            assert first.value == int(np.argmax(raw)), (nf, d_any)
