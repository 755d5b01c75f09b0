"""The drop-in glue (GeometricTrace + DeviceRows + dropin.accelerate) driven
end to end on CPU through a test double of the engine (tests/fake_engine.py,
oracle backed), against the unmodified reference on its own System objects
and its own ray generators."""
import numpy as np
import pytest

import rayopt_amd as ra
from rayopt_amd import dropin
from oracle import refshim

from fake_engine import OracleEngine

pytestmark = pytest.mark.skipif(not refshim.available(),
                                reason="no /root/reference")


@pytest.fixture()
def ro():
    mod = refshim.load()
    yield mod
    dropin.restore(mod)


def test_geometric_trace_on_reference_system(ro):
    s = ro.system_from_yaml(ra.prescriptions.TORTURE)
    y, u = ra.bundles.disc_bundle(500, 12., 3., 4)
    ref = ro.GeometricTrace(s)
    ref.rays_given(y[:, :2], u[:, :2])        # 2-component launch data
    mine = ra.GeometricTrace(s, engine=OracleEngine())
    mine.rays_given(y[:, :2], u[:, :2])
    for kw in (dict(clip=True), dict(clip=False), dict(clip=True, stop=-2)):
        with np.errstate(all="ignore"):
            ref.propagate(**kw)
        mine.propagate(**kw)
        b = range(len(s))[1:kw.get("stop")].stop
        for name in "yuit":
            assert np.array_equal(np.asarray(getattr(mine, name))[:b],
                                  getattr(ref, name)[:b], equal_nan=True)
        assert np.array_equal(mine.n[:b], ref.n[:b])
    for attr in ("path", "track", "origins", "mirrored"):
        assert np.array_equal(getattr(mine, attr), getattr(ref, attr))
    assert mine.y.shape == ref.y.shape and mine.t.shape == ref.t.shape
    assert np.array_equal(mine.y[-1, :, :2], ref.y[-1, :, :2], equal_nan=True)
    assert np.array_equal(mine.t[:4].sum(0), ref.t[:4].sum(0), equal_nan=True)


def test_accelerate_runs_reference_ray_generators(ro):
    """rays_point / rays_clipping / rays_line (aiming, pupils: rayopt's own
    host code) feeding the swapped-in propagate; reproduces the reference's
    known answer rms = 0.052 (test_raytrace.py:189-199) on numeric indices."""
    cls = dropin.accelerate(ro, engine_factory=OracleEngine)
    assert ro.GeometricTrace is cls and ro.analysis.GeometricTrace is cls
    text = ra.prescriptions.cooke().replace("radius: 20.", "radius: 0.364")
    s = ro.system_from_yaml(text)
    s.update()
    p = ro.ParaxialTrace(s)
    p.update_conjugates()
    g = ro.GeometricTrace(s)
    assert type(g).__mro__[1] is ra.GeometricTrace
    g.rays_point((0, 1.), nrays=13, distribution="radau", filter=False)
    np.testing.assert_allclose(g.rms(), .052, rtol=1e-2)
    ref = cls._reference_class(s)
    ref.rays_point((0, 1.), nrays=13, distribution="radau", filter=False)
    assert np.array_equal(np.asarray(g.y), ref.y, equal_nan=True)
    assert g.rms() == pytest.approx(ref.rms(), rel=1e-14)
    g.rays_clipping((0, 1.))
    g.rays_line((0, 1.))
    ref.rays_line((0, 1.))
    assert np.array_equal(np.asarray(g.u), ref.u, equal_nan=True)
    d0 = float(s[-1].distance)
    g.rays_point((0, 0.), nrays=21, distribution="hexapolar")
    shift = g.refocus()
    d1 = float(s[-1].distance)
    assert d1 == pytest.approx(d0 + shift) and abs(shift) < 1.
    s[-1].distance = d0
    ref.rays_point((0, 0.), nrays=21, distribution="hexapolar")
    ref.refocus()
    assert float(s[-1].distance) == pytest.approx(d1, rel=1e-12)
    assert np.array_equal(np.asarray(g.y), ref.y, equal_nan=True)
    dropin.restore(ro)
    assert ro.GeometricTrace is cls._reference_class


def test_keep_rows_glue(ro):
    s = ro.system_from_yaml(ra.prescriptions.DOUBLE_GAUSS)
    y, u = ra.bundles.disc_bundle(100, 10., 0., 1)
    g = ra.GeometricTrace(s, engine=OracleEngine())
    g.rays_given(y, u)
    g.propagate(keep=[-1, 4])
    assert np.isfinite(g.y[-1]).all() and np.isfinite(g.y[4]).all()
    with pytest.raises(AssertionError):
        g.y[5]


def test_multi_wavelength_groups_glue(ro):
    """rays_given(l=[...]): one batch, one table per wavelength; equals the
    reference traced once per wavelength (catalogue-free Abbe glasses)."""
    text = ra.prescriptions.SINGLET.replace("material: 1.5168",
                                            "material: 1.5168/64.17")
    s = ro.system_from_yaml(text)
    mine = ra.system_from_yaml(text)
    y, u = ra.bundles.disc_bundle(128, 7., 2., 3)
    ls = [486.13e-9, 587.56e-9, 656.27e-9]
    g = ra.GeometricTrace(mine, engine=OracleEngine())
    g.rays_given(y, u, ls)
    g.propagate(clip=True)
    assert g.y.shape == (4, 384, 3) and g.n.shape == (3, 4)
    for k, l in enumerate(ls):
        ref = ro.GeometricTrace(s)
        ref.rays_given(y, u, l)
        with np.errstate(all="ignore"):
            ref.propagate(clip=True)
        sl = slice(k*128, (k + 1)*128)
        for name in "yui":
            assert np.array_equal(np.asarray(getattr(g, name))[:, sl],
                                  getattr(ref, name), equal_nan=True)
        assert np.array_equal(np.asarray(g.t)[:, sl], ref.t, equal_nan=True)
        assert np.array_equal(g.n[k], ref.n)
    assert g.n[0, 1] != g.n[2, 1]
    with pytest.raises(ValueError, match="multiple of 64"):
        g.rays_given(y[:100], u[:100], ls)


def _run_analysis(ro, engine_factory, aim=True):
    """rayopt's own top-level consumer (rayopt/analysis.py:78-146: refocus,
    ray fans, spots at five defocus positions, OPD + PSF + encircled energy
    per field, longitudinal aberrations) on the accelerated trace."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    dropin.modernize(ro)
    dropin.accelerate(ro, engine_factory=engine_factory, aim=aim)
    s = ro.system_from_yaml(
        ra.prescriptions.cooke().replace("radius: 20.", "radius: 0.364"))
    s.update()
    before = s[-1].distance
    with np.errstate(all="ignore"):
        a = ro.Analysis(s, print=False)
    try:
        assert len(a.figures) == 5 and len(a.text) == 2
        assert "triplet" in a.text[0]
        # refocus_full moved the image plane by the least-squares shift
        assert 0 < abs(s[-1].distance - before) < 1.
    finally:
        for fig in a.figures:
            plt.close(fig)
    return s[-1].distance - before


@pytest.mark.parametrize("aim", [False, True])
def test_reference_analysis_runs_on_the_accelerated_trace(ro, aim):
    """Unmodified Analysis end to end through the swapped-in class (engine
    double on CPU); the refocus it performs equals the reference's own --
    exactly with rayopt's own aiming, to its aiming tolerance (1e-3) when
    System.pupil is answered by the aiming kernel."""
    shift = _run_analysis(ro, OracleEngine, aim)
    dropin.restore(ro)
    assert not getattr(ro.system.System.pupil, "_mi355", False)
    s = ro.system_from_yaml(
        ra.prescriptions.cooke().replace("radius: 20.", "radius: 0.364"))
    s.update()
    before = s[-1].distance
    t = ro.GeometricTrace(s)
    t.rays_point((0, 0.), nrays=13, distribution="radau", clip=False,
                 filter=False)
    t.refocus()
    assert shift == pytest.approx(s[-1].distance - before,
                                  rel=1e-3 if aim else 1e-9)


def test_device_pupil_behind_the_reference_interface(ro):
    """accelerate(aim=True): rayopt's System.pupil keeps its signature and
    return value, agrees with the reference's own solver to its tolerance,
    satisfies the aiming conditions far more tightly, and falls back to the
    reference's code for what the kernel does not model."""
    text = ra.prescriptions.cooke()
    s = ro.system_from_yaml(text)
    s.update()
    ro.ParaxialTrace(s).update_conjugates()
    want = [s.pupil(yo) for yo in ((0, 0.), (0, .7), (.3, -.4))]
    rim = s.pupil((0, 1.), stop=-1)
    dropin.accelerate(ro, engine_factory=OracleEngine)
    s2 = ro.system_from_yaml(text)
    s2.update()
    ro.ParaxialTrace(s2).update_conjugates()
    for yo, (z, a) in zip(((0, 0.), (0, .7), (.3, -.4)), want):
        z2, a2 = s2.pupil(yo)
        assert a2.shape == (2, 2)
        assert z2 == pytest.approx(z, rel=5e-3, abs=5e-3*6.25)
        np.testing.assert_allclose(a2, a, rtol=5e-3)
        assert s2.pupil(yo)[1] is not a2                 # cached, copied
    z2, a2 = s2.pupil((0, 1.), stop=-1)
    np.testing.assert_allclose(a2, rim[1], rtol=5e-3)
    # the chief ray of the aimed pupil goes through the stop centre
    g = ro.GeometricTrace(s2)
    g.rays((0, 1.), np.zeros((1, 2)), None, filter=False)
    assert np.abs(np.asarray(g.y[s2.stop])[0, :2]).max() < 1e-7
    # aiming switched off in the prescription: rayopt's own path answers
    s2.object.pupil.aim = False
    z3, a3 = s2.pupil((0, .5))
    assert z3 == s2.object.pupil.distance
    dropin.restore(ro)
    assert not getattr(ro.system.System.pupil, "_mi355", False)


@pytest.mark.gpu
def test_reference_analysis_runs_on_the_gpu(ro):
    _run_analysis(ro, None)


def test_modernize_is_idempotent_and_minimal():
    from matplotlib.axis import Axis
    dropin.modernize()
    dropin.modernize()
    assert np.int is int and np.complex_ is np.complex128
    assert hasattr(Axis, "set_smart_bounds")
    a = np.arange(6.).view(dropin.LegacyArray)
    assert a.ptp() == 5. and a[a > 2].ptp() == 2.
