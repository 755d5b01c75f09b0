"""aiming="reference": rayopt's aiming procedure (solvers, tolerances, guess
cache, call order; rayopt/system.py:466-593, rayopt/cachend.py:84-105) as
rayopt_amd/aiming_reference.py restates it, and aiming="rayopt": the
installed rayopt's own methods (rayopt_amd/dropin/aiming_rayopt.py), both with
the one-ray traces on the engine -- aimed pupils and the bundles launched
from them ARE the reference's, bit for bit, instead of agreeing to its 1e-3
(first-order start from the reference's own matrix products,
rayopt_amd/aiming.py: first_order_matrix).  What this proves: the engine's
TRACES under rayopt's solvers, and the restatement against the original.  It
is no evidence for this package's own aimer, FieldAimer
(tests/test_aiming.py: solver tolerance + defining conditions to 1e-9)."""
import numpy as np
import pytest

import rayopt_amd as ra
from rayopt_amd.aiming_reference import ReferenceAimer, reference_aimer
from rayopt_amd.prescriptions import cooke, ASPHERE_PHONE
from oracle import refshim

from fake_engine import OracleEngine

pytestmark = pytest.mark.skipif(not refshim.available(),
                                reason="no /root/reference")
COOKE = cooke().replace("radius: 20.", "radius: 0.364")
FINITE = COOKE.replace(
    "object: {angle_deg: 20, pupil: {radius: 6.25, aim: True}}",
    "object: {type: finite, radius: 8., pupil: {radius: 6.25, aim: True}}"
).replace("- {roc: 21.25, distance: 5.0,", "- {roc: 21.25, distance: 60.,")


def both(text):
    ro = refshim.load()
    rs = ro.system_from_yaml(text)
    rs.update()
    ms = ra.system_from_yaml(text)
    ms.update()
    return ro, rs, ms


KINDS = ("reference", "rayopt")


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("text", [COOKE, FINITE])
@pytest.mark.parametrize("stop", [None, -1])
def test_pupils_match_the_reference(text, stop, kind):
    ro, rs, ms = both(text)
    aimer = reference_aimer(ms, OracleEngine(), ms.wavelengths[0], stop,
                            None, kind)
    # the order matters: every field is seeded from the ones before it
    for yo in ((0, 1.), (0, .7), (0, 0.), (.6, .8), (0, .35), (0, 1.)):
        zr, ar = rs.pupil(yo, stop=stop)
        zm, am = aimer.pupil(yo)
        # bit for bit: the same solvers on the same one-ray traces from the
        # same first-order start (rayopt_amd/aiming.py: first_order_matrix)
        assert zm == zr, (yo, zm - zr)
        assert np.array_equal(am, ar), yo
    assert aimer.evaluations > 50


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("text", [COOKE, FINITE])
def test_generators_match_the_reference(text, kind):
    ro, rs, ms = both(text)
    r = ro.GeometricTrace(rs)
    g = ra.GeometricTrace(ms, engine=OracleEngine(), aiming=kind)
    for kind, args, kw in (
            ("rays_point", ((0, 1.),),
             dict(nrays=21, distribution="hexapolar", filter=False)),
            ("rays_point", ((0, .7),), dict(nrays=30, distribution="tee",
                                            clip=True)),
            ("rays_clipping", ((0, 1.),), {}),
            ("rays_line", ((0, 1.),), dict(nrays=5)),
            ("rays_point", ((0, 0.),), dict(nrays=13, distribution="square"))):
        getattr(r, kind)(*args, **kw)
        getattr(g, kind)(*args, **kw)
        assert g.nrays == r.nrays, kind
        for name in "yu":
            a, b = np.asarray(getattr(g, name)), getattr(r, name)
            assert np.array_equal(a, b, equal_nan=True), (kind, name)
    assert g.rms() == pytest.approx(r.rms(), rel=1e-12)


def test_update_forgets_the_guess_cache():
    ro, rs, ms = both(COOKE)
    g = ra.GeometricTrace(ms, engine=OracleEngine(), aiming="reference")
    g.rays_point((0, 1.), nrays=5)
    assert ms.__dict__["_reference_aimers"]
    ms[ms.stop].radius /= 2
    rs[rs.stop].radius /= 2
    ms.update()
    rs.update()
    assert "_reference_aimers" not in ms.__dict__
    g.rays_point((0, 1.), nrays=5)
    r = ro.GeometricTrace(rs)
    r.rays_point((0, 1.), nrays=5)
    assert np.array_equal(np.asarray(g.y[0]), r.y[0])


def test_aspheres_agree_through_the_newton_path():
    text = ASPHERE_PHONE.replace("pupil: {radius: 0.6}",
                                 "pupil: {radius: 0.6, aim: True}")
    ro, rs, ms = both(text)
    aimer = ReferenceAimer(ms, OracleEngine())
    for yo in ((0, 1.), (0, .5)):
        zr, ar = rs.pupil(yo)
        zm, am = aimer.pupil(yo)
        assert zm == zr and np.array_equal(am, ar)


def test_on_a_rayopt_system_the_cache_lives_in_its_pupil_cache():
    """GeometricTrace(rayopt_system, aiming="reference"): the solved fields
    are dropped when rayopt's own System.update() clears _pupil_cache."""
    ro, rs, ms = both(COOKE)
    g = ra.GeometricTrace(rs, engine=OracleEngine(), aiming="reference")
    r = ro.GeometricTrace(rs)
    g.rays_point((0, 1.), nrays=5)
    assert "rayopt_amd reference aimers" in rs._pupil_cache
    rs[rs.stop].radius /= 2
    rs.update()
    assert "rayopt_amd reference aimers" not in rs._pupil_cache
    g.rays_point((0, 1.), nrays=5)
    r.rays_point((0, 1.), nrays=5)
    assert np.array_equal(np.asarray(g.y[0]), r.y[0])
    assert np.array_equal(np.asarray(g.y[-1]), r.y[-1], equal_nan=True)


@pytest.mark.parametrize("text", [COOKE, FINITE])
def test_system_pupil_and_aim_methods(text):
    """System.pupil / System.aim of this package's System
    (rayopt/system.py:503-504, 585-593): launch rays for a given pupil equal
    the reference's to rounding; the aimed pupil to the reference's tolerance
    by the kernel, to 1e-11 by the reference procedure."""
    ro, rs, ms = both(text)
    eng = OracleEngine()
    yp = np.array([(0, 0), (0, .5), (.5, 0), (-.7, .7), (0, -1.), (.9, .9)])
    for yo in ((0, 1.), (.6, .8)):
        zr, ar = rs.pupil(yo)
        for filt in (True, False):
            yr, ur = rs.aim(yo, yp, zr, ar, filter=filt)
            ym, um = ms.aim(yo, yp, zr, ar, filter=filt, engine=eng)
            assert ym.shape == yr.shape
            np.testing.assert_allclose(ym, yr, rtol=0, atol=1e-12)
            np.testing.assert_allclose(um, ur, rtol=0, atol=1e-12)
        y1, u1 = ms.aim(yo, None, zr, ar, engine=eng)          # chief ray
        yr1, ur1 = rs.aim(yo, None, zr, ar)
        np.testing.assert_allclose(np.c_[y1, u1], np.c_[yr1, ur1],
                                   atol=1e-12)
        zm, am = ms.pupil(yo, engine=eng)
        assert zm == pytest.approx(zr, rel=3e-3)
        np.testing.assert_allclose(am, ar, rtol=3e-3)
        zm, am = ms.pupil(yo, aiming="reference", engine=OracleEngine())
        assert zm == zr and np.array_equal(am, ar)
    assert ra.FullTrace is ra.GeometricTrace


TILTED = COOKE.replace("- {roc: 21.25, distance: 5.0,",
                       "- {roc: 21.25, distance: 5.0, angles: [0.02, -0.01, 0],")


@pytest.mark.parametrize("kind", KINDS)
def test_tilted_elements_to_rounding(kind):
    """The solvers' traces are ONE-ray traces, and numpy hands a (1,3) @ (3,3)
    product to another BLAS routine (gemv) than an (N,3) one (gemm): the
    rotated vectors of a single ray are not the bits the kernel's FMA chain
    (and the double) produce.  Aimed pupils of a system with a tilted element
    therefore equal the reference's to rounding, not bit for bit
    (INTEGRATION.md section 1)."""
    assert "angles" in TILTED
    ro, rs, ms = both(TILTED)
    aimer = reference_aimer(ms, OracleEngine(), ms.wavelengths[0], None, None,
                            kind)
    for yo in ((0, 1.), (0, .7), (.6, .8)):
        zr, ar = rs.pupil(yo)
        zm, am = aimer.pupil(yo)
        assert zm == pytest.approx(zr, rel=1e-10, abs=1e-10)
        np.testing.assert_allclose(am, ar, rtol=1e-10, atol=1e-10)


def test_without_rayopt_the_restatement_works_and_the_binding_says_so():
    """aiming="reference" needs scipy and this package only; aiming="rayopt"
    raises ImportError when rayopt is not importable -- and nothing is ever
    written into the rayopt module."""
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
import rayopt_amd as ra
from rayopt_amd.prescriptions import cooke
from fake_engine import OracleEngine
assert "rayopt" not in sys.modules
ms = ra.system_from_yaml(cooke().replace("radius: 20.", "radius: 0.364"))
ms.update()
z, a = ms.pupil((0, 1.), aiming="reference", engine=OracleEngine())
assert np.isfinite(z) and np.isfinite(a).all()
try:
    ms.pupil((0, .5), aiming="rayopt", engine=OracleEngine())
except ImportError as err:
    assert "aiming='reference'" in str(err)
else:
    raise SystemExit("aiming='rayopt' did not raise without rayopt")
assert "rayopt" not in sys.modules
print("ok")
""" % (root, os.path.join(root, "tests"))
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    res = subprocess.run([sys.executable, "-c", code], capture_output=True,
                         text=True, env=env, cwd="/")
    assert res.returncode == 0 and "ok" in res.stdout, res.stderr[-1500:]
    ro = refshim.load()
    both(COOKE)
    reference_aimer(ra.system_from_yaml(COOKE), OracleEngine(), 587.56e-9,
                    None, None, "rayopt")
    assert not [k for k in vars(ro) if "mi355" in k.lower()]
