"""CPU (not gpu): the two claims the coarse pass on bit planes rests on, checked against the ORACLE's `similarity`
(line2Dup.cpp:807-858, :924-984) with numpy bit counts -- no GPU, no product code:

  (1) exactness: with any(o, p) = (LM[o][p] > 0) and exact(o, p) = (LM[o][p] == 4), read at the feature's FLAT offset (row
      overrun and zero tail included), the reference's sum over a template's features is  3 * #any + #exact;
  (2) the pruning bound: raw <= 4 * (nf - misses) where misses counts the features whose any-bit is 0 -- so a position with
      more than floor((4 nf - rmin) / 4) misses can never reach rmin, after ANY prefix of the features.

The GPU kernels (csrc/sbm_coarse_bits.h) implement exactly this arithmetic; their parity tests are tests/test_gpu_coarse_*.py."""
import numpy as np

from shape_based_matching_amd import synth


def flat_offsets(level_rec, feats, T, W, H, lm_stride):
    """per in-image feature: (orientation plane, flat offset inside it) = accessLinearMemory (:782-805)"""
    f = feats[int(level_rec["feature_offset"]): int(level_rec["feature_offset"]) + int(level_rec["n_features"])]
    out = []
    for x, y, lab in zip(f["x"].tolist(), f["y"].tolist(), f["label"].tolist()):
        if 0 <= x < W * T and 0 <= y < H * T:
            out.append((lab, ((y % T) * T + x % T) * W * H + (y // T) * W + x // T))
    return out


def test_sum_of_two_bits_is_the_reference_sum_and_misses_bound_it(oracle):
    for seed, rows, cols, T, nf, box in ((3, 256, 384, (4, 8), [40, 24], 64), (4, 320, 320, (8, 8), [90, 70], 80), (5, 192, 256, (4,), [63], 48)):
        L = len(T)
        maps, ts = synth.stage_b(seed, rows, cols, T, 12, nf, templ_size=box, plant_every=3, density_permille=40)
        pyr = oracle.Pyramid.from_quantized(maps, T)
        lc = L - 1
        Tc = T[lc]
        W, H = (cols >> lc) // Tc, (rows >> lc) // Tc
        lm = pyr.lm(lc)  # [8][lm_stride] response bytes, zero tail
        any_b, exact_b = lm > 0, lm == 4
        for t in range(ts.n_templates):
            rec = ts.levels[t, lc]
            want = pyr.similarity(rec, ts.features, lc).astype(np.int64).ravel()  # H x W score map of the oracle
            wf, hf = (int(rec["width"]) - 1) // Tc + 1, (int(rec["height"]) - 1) // Tc + 1
            npos = (H - hf) * W + (W - wf) + 1
            if npos <= 0:
                assert not want.any()
                continue
            offs = flat_offsets(rec, ts.features, Tc, W, H, lm.shape[1])
            n_all = int(rec["n_features"])
            a = np.zeros(npos, np.int64)
            e = np.zeros(npos, np.int64)
            misses = np.full(npos, n_all - len(offs), np.int64)  # features outside the image add nothing: misses from the start
            for k, (o, off) in enumerate(offs):
                ab = any_b[o, off: off + npos]
                a += ab
                e += exact_b[o, off: off + npos]
                misses += ~ab
                if k % 7 == 0:  # (2) after any prefix: what is still to come adds at most 4 per feature
                    assert (want[:npos] <= 4 * (n_all - misses)).all()
            assert np.array_equal(3 * a + e, want[:npos]), (seed, t)      # (1)
            assert (want[:npos] <= 4 * (n_all - misses)).all()            # (2) at the end
            assert not want[npos:].any()                                  # positions past the span score 0
        pyr.free()
