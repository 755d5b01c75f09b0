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


def test_sequences_on_the_shipped_asphere_arithmetic():
    """A slice of the same fuzz (> 2000 steps) with the DEFAULT arithmetic for
    even aspheres: aspheric systems within the 1e-8 contract of the double
    with identical NaN masks after every step, the others bit for bit."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RT_FUZZ_ARITH="default",
               RT_MI355_EXACT_ASPHERE="0")
    out = subprocess.run(
        [sys.executable, os.path.join(root, "tests", "tools", "fuzz_state.py"),
         "9000", "9110"], env=env, capture_output=True, text=True,
        timeout=600)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    last = out.stdout.strip().splitlines()[-1]
    assert last.startswith("state fuzz") and last.endswith(
        " 0 failing sequences"), out.stdout[-2000:]
    import re
    assert int(re.search(r"(\d+) steps", last).group(1)) > 2000, last
