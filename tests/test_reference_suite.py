"""The reference's own geometric-trace test case, restated against this
package: rayopt/test/test_raytrace.py:60-199 (``DemotripCase``: the Cooke
fixture by glass name, with its pickups and validators), the parts that lie on
the traced path or feed it.  Same assertions, same tolerances.  Runs on the
engine double here and on the device under ``-m gpu``.  (``setUp`` there also
calls ``paraxial.refocus()``; the paraxial trace is outside this package and
the fixture's image plane is 0.02 mm from that focus -- none of the
assertions depends on it.)"""
import numpy as np
import numpy.testing as nptest
import pytest

import rayopt_amd as ra
from test_design import COOKE


def cpu_engine():
    from fake_engine import OracleEngine
    return OracleEngine()


ENGINES = [pytest.param(cpu_engine, id="engine-double"),
           pytest.param(lambda: None, id="device", marks=pytest.mark.gpu)]


@pytest.fixture
def s():
    system = ra.system_from_yaml(COOKE)
    system.update()
    return system


def check_from_text(s):
    assert not s.object.finite
    for i, el in enumerate(s):
        if i not in (0,):
            assert el.radius > 0
        if i not in (0, s.stop):
            assert el.distance > 0
        if i not in (0, s.stop, len(s) - 1):
            assert abs(el.curvature) > 0
        if i not in (len(s) - 1,):
            assert el.material is not None


def check_system(s):
    assert len(str(s).splitlines()) > 10
    assert s.aperture is s[s.stop]


def test_from_text_and_system(s):
    check_from_text(s)
    check_system(s)


def test_reverse(s):
    before = s.dict()
    s.reverse()
    assert s.object.finite and s[-1].material is None
    s.reverse()
    check_from_text(s)
    check_system(s)
    assert s.dict() == before


def test_rescale(s):
    l = [el.distance for el in s]
    s.rescale(123)
    nptest.assert_allclose([el.distance/123 for el in s], l)
    s.rescale()
    nptest.assert_allclose([el.distance for el in s], l)


def test_funcs(s):
    s.resize_convex()
    s.track
    s.origins
    s.mirrored
    s.align(np.ones_like(s.track))


@pytest.mark.parametrize("engine", ENGINES)
def test_aim_and_pupil(s, engine):
    g = ra.GeometricTrace(s, engine=engine())
    z, p = s.pupil((0, 1.)) if engine() is None else \
        g._pupil((0, 1.), s.wavelengths[0])
    assert np.isfinite(z).all() and np.isfinite(p).all()
    g.rays_point((0, 1.))
    g.rays_clipping((0, 1.))
    g.rays_line((0, 1.))
    for y in [(0, 0), (1, 0), (-1, 0), (0, 1), (0, -1), (.1, .1), (-.2, .5)]:
        z, a = g._pupil(y, s.wavelengths[0])
        assert np.isfinite(z).all() and np.isfinite(a).all()


@pytest.mark.parametrize("aiming", ["device", "reference"])
@pytest.mark.parametrize("engine", ENGINES)
def test_aim_point_more(s, engine, aiming):
    g = ra.GeometricTrace(s, engine=engine(), aiming=aiming)
    i = s.stop
    r = np.array([el.radius for el in s[1:-1]])

    g.rays_clipping((0, 1.))
    u0 = np.asarray(g.u[0])
    nptest.assert_allclose(u0, u0[(0,)*len(u0), :])
    y = np.asarray(g.y)
    nptest.assert_allclose(y[i, 0, 1], 0, atol=5e-3)
    nptest.assert_allclose(min(y[1:-1, 1, 1] + r), 0, atol=1e-3)
    nptest.assert_allclose(max(y[1:-1, 2, 1] - r), 0, atol=1e-3)

    g.rays_point((0, 1.), distribution="cross", nrays=5, filter=False)
    u0 = np.asarray(g.u[0])
    nptest.assert_allclose(u0, u0[(0,)*len(u0), :])
    y = np.asarray(g.y)
    nptest.assert_allclose(y[i, :3, 1]/s[i].radius, [-1, 0, 1], atol=1e-3,
                           rtol=3e-2)
    nptest.assert_allclose(y[i, :, 0]/s[i].radius, [0, 0, 0, -1, 0, 1],
                           atol=1e-1)
    g.rays_line((0, 1.))


@pytest.mark.parametrize("aiming", ["device", "reference"])
@pytest.mark.parametrize("engine", ENGINES)
def test_quadrature(s, engine, aiming):
    g = ra.GeometricTrace(s, engine=engine(), aiming=aiming)
    g.rays_point((0, 1.), nrays=13, distribution="radau", filter=False)
    a = g.rms()
    nptest.assert_allclose(a, .052, rtol=1e-2)
    g.rays_point((0, 1.), nrays=500, distribution="square", clip=False,
                 filter=True)
    b = g.rms()
    nptest.assert_allclose(a, b, rtol=5e-2)
