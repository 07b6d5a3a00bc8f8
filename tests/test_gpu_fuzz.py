"""-m gpu: a slice of tools/fuzz_match.py -- random frame geometry, channels, batch size, template subset, threshold,
mask, gradient-kernel mode and coarse-pass kernel through sbm_match_batch_device, every frame's match multiset against
the oracle (Detector::match, line2Dup.cpp:1078-1150)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_random_batches_against_oracle(oracle):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_match

    assert fuzz_match.run(24, 31, verbose=False) > 1000


def test_gradient_stage_fuzz_slice():
    """tools/fuzz_gradient.py: images built to sit on the gradient stage's decisions (threshold equality, channel ties, bin
    boundaries, vote ties, constant regions beside texture) x random geometry, pyramid, weak threshold, kernel and rows per
    work item; every byte of every level's orientation map against the oracle"""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_gradient

    assert fuzz_gradient.run(80, 17, verbose=False) > 5e6
