"""The SHIPPED default for even aspheres: a context created without
RT_MI355_EXACT_ASPHERE (tests/conftest.py sets it for the bit-identity suite)
and without options runs the FMA / rcp / rsq Newton solve (rt_math.h:
rt_newton_fast) -- same iteration as rayopt/elements.py:333-349 (x0 = plane
intercept, |step| <= 1e-7, five iterates, NaN on failure), results within the
1e-8 contract of iterated aspheres against the reference's own goldens with
identical NaN masks; ``exact_asphere=True`` gives the reference's bits."""
import numpy as np
import pytest

import rayopt_amd as ra
from rayopt_amd.pack import resolve_range
from conftest import golden_names, load_golden, assert_parity, RTOL_ASPHERE

pytestmark = pytest.mark.gpu

ASPHERIC = [n for n in golden_names() if "aspherics" in load_golden(n)["yaml"]]


def trace(gold, monkeypatch, **options):
    monkeypatch.delenv("RT_MI355_EXACT_ASPHERE", raising=False)
    system = ra.system_from_yaml(gold["yaml"])
    g = ra.GeometricTrace(system, **options)
    g.rays_given(gold["y0"], gold["u0"], gold["l"])
    g.propagate(start=gold["start"], stop=gold["stop"], clip=gold["clip"])
    a, b = resolve_range(len(system), gold["start"], gold["stop"])
    return ([np.array(np.asarray(rows[a:b])) for rows in (g.y, g.u, g.i, g.t)],
            [gold[k][a:b] for k in "yuit"])


@pytest.mark.parametrize("name", ASPHERIC)
def test_default_context_meets_the_asphere_contract(name, monkeypatch):
    gold = load_golden(name)
    got, want = trace(gold, monkeypatch)
    for k, rows, ref in zip("yuit", got, want):
        assert_parity(rows, ref, RTOL_ASPHERE, "%s %s" % (name, k))
    # ... and it IS the fast arithmetic: the bits of fast_asphere=1
    fast, _ = trace(gold, monkeypatch, fast_asphere=1)
    for a, b in zip(got, fast):
        assert np.array_equal(a, b, equal_nan=True)
    # exact_asphere=True: the reference's bits (tilted systems: where this
    # host's BLAS follows the FMA chain, tests/conftest.py)
    exact, _ = trace(gold, monkeypatch, exact_asphere=True)
    for k, rows, ref in zip("yuit", exact, want):
        assert_parity(rows, ref, 1e-13, "%s exact %s" % (name, k))


def test_the_two_arithmetics_are_not_the_same_bits(monkeypatch):
    """Guard against the option silently doing nothing."""
    gold = load_golden("asphere_12")
    a, _ = trace(gold, monkeypatch)
    b, _ = trace(gold, monkeypatch, exact_asphere=True)
    assert any(not np.array_equal(x, w, equal_nan=True)
               for x, w in zip(a, b))
    for x, w in zip(a, b):
        assert np.array_equal(np.isnan(x), np.isnan(w))
