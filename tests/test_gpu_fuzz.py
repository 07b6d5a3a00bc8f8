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
