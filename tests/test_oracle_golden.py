"""The oracle (oracle/trace_numpy.py) against the reference's outputs.

Golden vectors were produced by the unmodified reference
(tests/golden/make_golden.py).  Sphere/conic/plane paths must reproduce them
bit for bit (the oracle performs the same numpy operations); the asphere
path (vectorised restatement of scipy's scalar Newton) to 1e-12.
"""
import numpy as np
import pytest

import rayopt_amd as ra
from rayopt_amd.pack import pack_system, resolve_range
from oracle import trace_numpy as tn
from oracle import refshim

from conftest import golden_names, load_golden, assert_parity


def oracle_on_case(g):
    system = ra.system_from_yaml(g["yaml"])
    a, b = resolve_range(len(system), g["start"], g["stop"])
    n0 = system.refractive_index(g["l"], 0)
    table, ns = pack_system(system, g["l"], n0, a, b)
    Y, U, I, T = tn.propagate(table, g["y0"], g["u0"], a, b, g["clip"])
    return system, a, b, ns, (Y, U, I, T)


@pytest.mark.parametrize("name", golden_names())
def test_oracle_matches_reference_golden(name):
    g = load_golden(name)
    system, a, b, ns, (Y, U, I, T) = oracle_on_case(g)
    # bit-exact, all of it: tilted and decentred elements (the host model
    # forms rot_normal with the reference's own products,
    # rayopt_amd/model.py _euler_rxyz / _axis_angle) and the iterated
    # aspheres (scipy's scalar Newton restated operation for operation, its
    # derivative summed with the fused chain the reference reaches through
    # BLAS, oracle/trace_numpy.py: fma)
    for label, got, want in (("y", Y, g["y"]), ("u", U, g["u"]),
                             ("i", I, g["i"]), ("t", T, g["t"])):
        assert np.array_equal(got, want[a:b], equal_nan=True), (name, label)
    assert np.array_equal(ns[a:b], g["n"][a:b])
    # rows the reference did not trace were poisoned by the generator
    assert np.isnan(g["t"][b:]).all()


def test_reference_known_answer_rms():
    """rayopt/test/test_raytrace.py:189-199: rms == 0.052 (rtol 1e-2) for 13
    Radau rays at field (0,1) of the Cooke fixture.  The launch rays and the
    per-surface catalogue indices come from the reference run; geometry is
    this package's COOKE prescription."""
    with np.load("tests/golden/kat_cooke_quadrature.npz") as z:
        k = {key: z[key] for key in z.files}
    system = ra.system_from_yaml(ra.prescriptions.cooke())
    # the prescription restates the fixture geometry (image radius aside)
    assert np.allclose([e.distance for e in system], k["distance"])
    assert np.allclose([e.curvature for e in system], k["curvature"])
    # use exactly the indices the reference's catalogue produced
    for el, n in zip(system, k["n"]):
        if el.material is not None:
            el.material = ra.ConstantIndex(float(n))
    table, ns = pack_system(system, float(k["l"]), float(k["n"][0]))
    Y, U, I, T = tn.propagate(table, k["y0"], k["u0"], clip=False)
    assert np.array_equal(Y, k["y"][1:], equal_nan=True)
    y = Y[-1][:, :2]
    r = np.square(y - y.mean(0)).sum(1)
    rms = np.sqrt((r*k["w"]).sum())        # GeometricTrace.rms, :171-183
    assert rms == pytest.approx(float(k["rms"]), rel=1e-14)
    np.testing.assert_allclose(rms, .052, rtol=1e-2)   # the reference's pin


def test_survey_anchor_values():
    """Golden anchors quoted in SURVEY.md section 8c were for indices
    1.0/1.62041/1.62005; here only the structure (3 rays, all finite, total
    path ~68) is sanity-checked and the stored reference arrays are the
    pin."""
    g = load_golden("cooke_anchor")
    assert g["y"].shape == (9, 3, 3)
    assert np.isfinite(g["y"]).all()
    assert np.allclose(g["t"].sum(0), 68.5, atol=1.)


@pytest.mark.skipif(not refshim.available(), reason="no /root/reference")
@pytest.mark.parametrize("key", sorted(ra.prescriptions.ALL))
def test_oracle_matches_live_reference(key):
    """Fresh seeds, directly against the in-place imported reference."""
    ro = refshim.load()
    text = ra.prescriptions.ALL[key]
    rng = np.random.default_rng(sum(map(ord, key)))
    ref_sys = ro.system_from_yaml(text)
    rad = min(float(e.radius) for e in ref_sys[1:-1])*1.15
    n = 64 if "aspherics" in text else 2000
    y, u = ra.bundles.disc_bundle(n, rad, float(rng.uniform(0, 3)),
                                  int(rng.integers(1 << 30)))
    for clip in (True, False):
        g = ro.GeometricTrace(ref_sys)
        g.rays_given(y, u)
        with np.errstate(all="ignore"):
            g.propagate(clip=clip)
        # the packer accepts the reference's own System object
        table, ns = pack_system(ref_sys, g.l, g.n[0])
        Y, U, I, T = tn.propagate(table, g.y[0], g.u[0], clip=clip)
        tol = 1e-12 if "aspherics" in text else 0.
        for got, want in ((Y, g.y[1:]), (U, g.u[1:]), (I, g.i[1:]),
                          (T, g.t[1:])):
            if tol:
                assert_parity(got, want, tol, key)
            else:
                assert np.array_equal(got, want, equal_nan=True)
        assert np.array_equal(ns[1:], g.n[1:])
        # and this package's model packs to the same table
        mine, _ = pack_system(ra.system_from_yaml(text), g.l, g.n[0])
        for field in mine.dtype.names:
            np.testing.assert_allclose(mine[field], table[field], rtol=0,
                                       atol=1e-15)


NEGATIVE_INDEX = """
wavelengths: [587.56e-9]
elements:
- {material: 1.0}
- {roc: 40, distance: 10, material: -1.5, radius: 12}
- {roc: -60, conic: -0.5, distance: 6, material: 1.2, radius: 12}
- {distance: 30, radius: 50}
"""


@pytest.mark.skipif(not refshim.available(), reason="no /root/reference")
def test_negative_index_medium_matches_reference():
    """mu < 0 but != -1: refraction with sign(mu) = -1
    (rayopt/elements.py:365-367), not a mirror."""
    ro = refshim.load()
    s = ro.system_from_yaml(NEGATIVE_INDEX)
    y, u = ra.bundles.disc_bundle(500, 10., 3., 1)
    g = ro.GeometricTrace(s)
    g.rays_given(y, u)
    with np.errstate(all="ignore"):
        g.propagate(clip=True)
    table, ns = pack_system(ra.system_from_yaml(NEGATIVE_INDEX), g.l, g.n[0])
    assert table["mu"][1] < 0 and table["smu"][1] == -1 and \
        not table["flags"][1] & 0x40
    got = tn.propagate(table, y, u, clip=True)
    for a, b in zip(got, (g.y[1:], g.u[1:], g.i[1:], g.t[1:])):
        assert np.array_equal(a, b, equal_nan=True)
    assert np.isfinite(g.y[-1]).any()
