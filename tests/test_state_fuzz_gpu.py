"""A slice of the stateful fuzz (tests/tools/fuzz_state.py) in the GPU suite:
random sequences of seedings, partial traces, kept-row subsets, row uploads,
reads, reductions, wavelength groups, system variants, aimed polychromatic
bundles and kernel options on the device against the same sequence on the
numpy engine double -- every held row bit for bit after every step."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "tools"))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("block", range(5))
def test_random_call_sequences_against_the_engine_double(block):
    """5 x 17 sequences x 30 operations (> 2000 steps incl. in-place edits of
    element arrays and chunked traces), every held row checked after every
    step."""
    import fuzz_state
    steps = 0
    for seed in range(5000 + 17*block, 5000 + 17*(block + 1)):
        with np.errstate(all="ignore"):
            steps += fuzz_state.sequence(seed, 30)
    assert steps > 200
