"""The rest of the reference's Trace / GeometricTrace interface
(rayopt/raytrace.py:38-67, rayopt/geometric_trace.py:242-259): from_axis,
print_coeffs, align, print_trace / text / str -- against the reference on the
same systems and rays (engine double on CPU)."""
import numpy as np
import pytest

import rayopt_amd as ra
from rayopt_amd.prescriptions import TORTURE, cooke
from oracle import refshim

from fake_engine import OracleEngine

pytestmark = pytest.mark.skipif(not refshim.available(),
                                reason="no /root/reference")


def pair(text, n=7):
    ro = refshim.load()
    rs = ro.system_from_yaml(text)
    ms = ra.system_from_yaml(text)
    y, u = ra.bundles.disc_bundle(n, 3., 1., 2)
    r = ro.GeometricTrace(rs)
    g = ra.GeometricTrace(ms, engine=OracleEngine())
    for t in (r, g):
        t.rays_given(y, u)
        with np.errstate(all="ignore"):
            t.propagate()
    return r, g


@pytest.mark.parametrize("text", [cooke(), TORTURE])
def test_text_is_the_reference_text(text):
    r, g = pair(text)
    assert str(g) == str(r)
    assert list(g.text()) == list(r.text())
    assert list(g.print_trace(rays=[2])) == \
        list(r.print_trace())[2*(len(r.system) + 3):3*(len(r.system) + 3)]
    c = np.arange(3.*len(r.system)).reshape(-1, 3)
    for total in (True, False):
        assert list(g.print_coeffs(c, "abc", sum=total)) == \
            list(r.print_coeffs(c, "abc", sum=total))


def test_from_axis():
    r, g = pair(TORTURE)
    rng = np.random.default_rng(0)
    pts = rng.uniform(-1, 1, (40, 5, 3))
    pts[..., 2] = np.sort(rng.uniform(-5, r.path[-1] + 20, (40, 1)), axis=0)
    want = r.from_axis(pts.copy())
    got = g.from_axis(pts.copy())
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)
    np.testing.assert_allclose(g.from_axis(pts[:, 0], i=[3, 9, 9, 20]),
                               r.from_axis(pts[:, 0], i=[3, 9, 9, 20]),
                               atol=1e-12)


def test_align_tilts_the_elements_like_the_reference():
    r, g = pair(TORTURE)
    with np.errstate(all="ignore"):
        r.align()
    g.align()
    for a, b in zip(g.system, r.system):
        np.testing.assert_allclose(a.angles, b.angles, atol=1e-12)
        assert a.rotated == b.rotated
        if b.rotated:
            np.testing.assert_allclose(a.rot_normal, b.rot_normal, atol=1e-12)
    for name in "yut":
        a, b = np.asarray(getattr(g, name)), getattr(r, name)
        assert np.array_equal(np.isnan(a), np.isnan(b))
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-9, equal_nan=True)
