"""The opt-in fast arithmetic for even aspheres (RT_F_FAST, rt_set_option
"fast_asphere"): same Newton iteration as the reference's
(rayopt/elements.py:333-349 + scipy newton: x0 = plane intercept,
|step| <= 1e-7, five iterates, NaN on failure) on FMA / rcp / rsq arithmetic
with one reciprocal per iterate.  Contract: 1e-8 relative for iterated
aspheres (BASELINE north_star); asserted here two orders tighter, with
identical NaN masks, against the reference's golden outputs.

CPU: the kernel's arithmetic header compiled for the host (the hardware
rcp/rsq seeds are replaced by exact ones there, the refinement and the
algebra are the same).  GPU: through the C ABI, `-m gpu`."""
import numpy as np
import pytest

import rayopt_amd as ra
from rayopt_amd._lib import F_ASPH, F_FAST
from rayopt_amd.pack import pack_system, resolve_range

from conftest import golden_names, load_golden, assert_parity

FAST_RTOL = 1e-10
ASPHERE_GOLDENS = [n for n in golden_names()
                   if "aspherics" in load_golden(n)["yaml"]]


def fast_table(table):
    t = table.copy()
    t["flags"] = np.where(t["flags"] & F_ASPH, t["flags"] | F_FAST,
                          t["flags"])
    return t


def test_there_are_asphere_goldens():
    assert len(ASPHERE_GOLDENS) >= 4


@pytest.mark.parametrize("rays_per_lane", [1, 2])
@pytest.mark.parametrize("name", ASPHERE_GOLDENS)
def test_fast_arithmetic_on_the_host_build(hostemu, name, rays_per_lane):
    g = load_golden(name)
    system = ra.system_from_yaml(g["yaml"])
    a, b = resolve_range(len(system), g["start"], g["stop"])
    table, ns = pack_system(system, g["l"],
                            system.refractive_index(g["l"], 0), a, b)
    Y, U, I, T = hostemu(fast_table(table), g["y0"], g["u0"], a, b,
                         g["clip"], rays_per_lane)
    for label, got, want in (("y", Y, g["y"]), ("u", U, g["u"]),
                             ("i", I, g["i"]), ("t", T, g["t"])):
        assert_parity(got, want[a:b], FAST_RTOL, "%s.%s" % (name, label))


def test_fast_path_handles_many_terms_and_flat_base(hostemu):
    """Term-count classes (<= 4, <= 7, <= 10) and an asphere without base
    curvature, against the exact path of the same header."""
    rng = np.random.default_rng(5)
    for nterm, roc in ((2, 30.), (6, -25.), (10, 40.), (3, 0.)):
        coef = (rng.standard_normal(nterm)*10.**(-3 - 2*np.arange(nterm)))
        text = """
object: {type: infinite, angle_deg: 3}
elements:
- {material: air, radius: 10}
- {%s conic: %s, aspherics: %s, distance: 5, material: 1.6, radius: 9}
- {%s aspherics: %s, distance: 4, material: 1.0, radius: 9}
- {distance: 30, radius: 30}
""" % ("roc: %g," % roc if roc else "", -0.7 if roc else 0.,
            [float(c) for c in coef], "roc: %g," % (-2*roc) if roc else "",
            [float(-c) for c in coef])
        system = ra.system_from_yaml(text)
        y = np.zeros((4000, 3))
        y[:, :2] = rng.uniform(-8.5, 8.5, (4000, 2))
        u = np.zeros((4000, 3))
        u[:, :2] = rng.uniform(-.05, .05, (4000, 2))
        u[:, 2] = np.sqrt(1 - np.square(u[:, :2]).sum(1))
        table, ns = pack_system(system, .5876, 1.)
        L = len(system)
        exact = hostemu(table, y, u, 1, L, True, 1)
        fast = hostemu(fast_table(table), y, u, 1, L, True, 1)
        assert np.isfinite(exact[0][-1]).mean() > .5
        for label, got, want in zip("yuit", fast, exact):
            assert_parity(got, want, FAST_RTOL, "nterm%d.%s" % (nterm, label))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ASPHERE_GOLDENS)
def test_fast_asphere_on_the_device(name):
    g = load_golden(name)
    system = ra.system_from_yaml(g["yaml"])
    a, b = resolve_range(len(system), g["start"], g["stop"])
    tr = ra.GeometricTrace(system)
    tr.engine.set_option("fast_asphere", 1)
    try:
        tr.rays_given(g["y0"], g["u0"], g["l"])
        if a > 1:
            pytest.skip("partial golden: seeded rows are exact by definition")
        tr.propagate(a, b, clip=g["clip"])
        for label, rows in (("y", tr.y), ("u", tr.u), ("i", tr.i),
                            ("t", tr.t)):
            assert_parity(np.asarray(rows[a:b]), g[label][a:b], FAST_RTOL,
                          "%s.%s" % (name, label))
    finally:
        tr.engine.set_option("fast_asphere", 0)


@pytest.mark.gpu
def test_fast_asphere_c4_against_the_exact_path():
    """BASELINE configs[3] (six even aspheres) at 2e6 rays: fast vs exact on
    the device, identical NaN masks, <= 1e-10."""
    from rayopt_amd import prescriptions as P
    system = ra.system_from_yaml(P.ASPHERE_PHONE)
    deg = 17.5                  # as scripts/configs_bench.py builds C4
    y, u = ra.bundles.disc_bundle(2_000_000, 0.6, deg, 3)
    y[:, 1] -= 0.5*np.tan(np.radians(deg))
    tr = ra.GeometricTrace(system)
    L = len(system)
    tr.rays_given(y, u)
    tr.propagate(clip=True)
    exact = [np.asarray(r[L - 1]).copy() for r in (tr.y, tr.u, tr.t)]
    ex_all_t = np.asarray(tr.t[1:]).copy()
    tr.engine.set_option("fast_asphere", 1)
    try:
        tr.propagate(clip=True)
        fast = [np.asarray(r[L - 1]) for r in (tr.y, tr.u, tr.t)]
        assert np.isfinite(exact[0]).mean() > .5
        for label, got, want in zip(("y", "u", "t"), fast, exact):
            assert_parity(got[None], want[None], FAST_RTOL, "c4." + label)
        assert_parity(np.asarray(tr.t[1:]), ex_all_t, FAST_RTOL, "c4.t rows")
    finally:
        tr.engine.set_option("fast_asphere", 0)
