"""The device against the LIVE reference on the GPU box.

``oracle/make_ref.py`` (run by ``__graft_entry__.build()`` in the build
container) packs the unmodified reference into the git-ignored
``oracle/_ref/``, which travels to the GPU box with the snapshot; here
``rayopt.GeometricTrace.propagate`` (rayopt/geometric_trace.py:72-80) itself
runs on the box's host cores next to the engine, same System text, same rays,
and every value of y, u, i, t of every row is compared -- no oracle, no
recorded digest in between.  BASELINE configs C1, C2 (three wavelengths), C3
and C4 at sizes the reference finishes in seconds, plus the tilted / folded
torture system."""
import os
import sys

import numpy as np
import pytest

import rayopt_amd as ra
from oracle import refshim

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import digest_cases as dc  # noqa: E402
from conftest import assert_parity, blas_follows_fma_chain, RTOL_SPHERICAL

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not refshim.available(),
                                 reason="oracle/_ref was not built "
                                        "(python -m oracle.make_ref)")]

SUBSAMPLE = 100_000
CASES = {c["name"]: c for c in dc.cases(heavy=False)
         if "unclipped" not in c["name"]}


def reference_rows(case, y, u):
    ro = refshim.load()
    system = ro.system_from_yaml(case["yaml"])
    t = ro.GeometricTrace(system)
    t.rays_given(y, u, case["l"])
    with np.errstate(all="ignore"):
        t.propagate(clip=case["clip"])
    return t


@pytest.mark.parametrize("name", sorted(CASES))
def test_device_equals_the_live_reference(name):
    case = CASES[name]
    y, u = case["rays"]()
    y, u = y[:SUBSAMPLE], u[:SUBSAMPLE]
    want = reference_rows(case, y, u)
    system = ra.system_from_yaml(case["yaml"])
    g = ra.GeometricTrace(system)
    g.rays_given(y, u, case["l"])
    g.propagate(clip=case["clip"])
    tilted = any(getattr(e, "rotated", False) for e in system)
    exact = not tilted or blas_follows_fma_chain()
    for k in "yuit":
        got, ref = np.asarray(getattr(g, k)[:]), getattr(want, k)
        if exact:
            assert np.array_equal(got, ref, equal_nan=True), (name, k)
        else:   # this host's BLAS sums a 3-vector in another order
            assert_parity(got, ref, RTOL_SPHERICAL, "%s %s" % (name, k))
    assert np.array_equal(np.asarray(g.n), want.n)


def test_fast_asphere_default_meets_the_asphere_contract_vs_live_reference():
    """C4 on the DEFAULT arithmetic of aspheric elements (FMA / rcp / rsq
    Newton, rt_math.h) against the live reference: 1e-8 relative (BASELINE
    north_star), identical NaN masks; ``exact_asphere=True`` is the
    bit-identical path checked above."""
    case = CASES["C4_asphere_2e4_two_fields"]
    y, u = case["rays"]()
    want = reference_rows(case, y, u)
    system = ra.system_from_yaml(case["yaml"])
    g = ra.GeometricTrace(system, exact_asphere=False)
    g.rays_given(y, u, case["l"])
    g.propagate(clip=case["clip"])
    for k in "yuit":
        assert_parity(np.asarray(getattr(g, k)[:]), getattr(want, k), 1e-8,
                      "C4 fast %s" % k)
