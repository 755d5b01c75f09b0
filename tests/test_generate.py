"""On-device ray generation (rt_generate_rays): host frames + the generation
arithmetic (compiled for the host) against the reference's System.aim here;
the kernel against the oracle on the GPU."""
import numpy as np
import pytest

import rayopt_amd as ra
from rayopt_amd.launch import field_frames
from rayopt_amd.pack import pack_system
from oracle import aim_numpy as an
from oracle import trace_numpy as tn
from oracle import refshim

from conftest import assert_parity, RTOL_SPHERICAL

FINITE_OBJECT = ("object: {type: finite, radius: 25., "
                 "pupil: {radius: 16.0, distance: 120}}")
CURVED_OBJECT = "- {material: 1.0, roc: -300., conic: -0.4}\n"


def systems():
    base = ra.prescriptions.DOUBLE_GAUSS
    fin = base.replace("object: {angle_deg: 14, pupil: {radius: 16.0}}",
                       FINITE_OBJECT)
    out = {"infinite": base, "finite": fin,
           "finite_telecentric": fin.replace(
               "distance: 120}", "distance: 120, telecentric: true}"),
           "finite_curved_object": fin.replace("- {material: 1.0}\n",
                                               CURVED_OBJECT, 1),
           "infinite_curved_first": base.replace("- {material: 1.0}\n",
                                                 CURVED_OBJECT, 1)}
    for proj in ("stereographic", "equisolid", "orthographic", "equidistant"):
        out["infinite_" + proj] = base.replace(
            "object: {angle_deg: 14,",
            "object: {projection: %s, angle_deg: 34," % proj)
    return out


FIELDS = np.array([(0., 0.), (0., 1.), (0.3, -0.7), (-1., 0.2)])
ZA = [(68.94, 16.0), (-20., np.array(((-3., -5.), (3.5, 5.)))),
      (150., np.array(((-9., -9.), (9., 9.))))]


def pupil_points(n=200, seed=3):
    rng = np.random.default_rng(seed)
    return np.vstack([[(0., 0.)], rng.uniform(-1, 1, (n - 1, 2))])


def oracle_rays(system, yo, yp, z, a):
    obj = system.object
    if not obj.finite:
        return an.aim_infinite(obj.angle, yo, yp, z, a,
                               getattr(system[0], "curvature", 0.),
                               getattr(system[0], "conic", 0.),
                               obj.extra.get("projection", "rectilinear"))
    from rayopt_amd.launch import _sag0, _telecentric

    def sag(y):
        return np.array([_sag0(system[0], yi) for yi in y])
    return an.aim_finite(obj.radius, _telecentric(obj), yo, yp, z, a, sag)


@pytest.mark.skipif(not refshim.available(), reason="no /root/reference")
@pytest.mark.parametrize("key", sorted(systems()))
def test_frames_and_generation_math_vs_reference_aim(key, hostemu):
    ro = refshim.load()
    ref_sys = ro.system_from_yaml(systems()[key])
    mine = ra.system_from_yaml(systems()[key])
    yp = pupil_points()
    for z, a in ZA:
        a2 = a if np.ndim(a) else a*np.array(((-1., -1.), (1., 1.)))
        for s in (ref_sys, mine):     # frames from either kind of System
            frames = field_frames(s, FIELDS, z, a)
            table, _ = pack_system(s, 587.56e-9, 1.0)
            Y, U = hostemu.generate(frames, yp, table[:1])
            for f, yo in enumerate(FIELDS):
                yo_, uo_ = oracle_rays(mine, yo, yp, z, a2)
                try:
                    with np.errstate(all="ignore"):
                        yr, ur = ref_sys.aim(yo, yp, z, a2, filter=False)
                except ValueError:
                    # the reference's orthographic branch hstacks (n,2) with
                    # (n,1,1) (rayopt/conjugates.py:224-227) and raises: only
                    # the oracle's (intended) formula can be checked
                    assert "orthographic" in key
                    yr, ur = yo_, uo_
                assert np.array_equal(yr, yo_, equal_nan=True)
                assert np.array_equal(ur, uo_, equal_nan=True)
                sl = slice(f*len(yp), (f + 1)*len(yp))
                assert_parity(Y[sl][None], yr[None], 1e-13, key + ".y")
                assert_parity(U[sl][None], ur[None], 1e-13, key + ".u")


@pytest.mark.gpu
@pytest.mark.parametrize("key", sorted(systems()))
def test_device_generation_vs_oracle(key):
    system = ra.system_from_yaml(systems()[key])
    yp = pupil_points(5003, 9)
    z, a = ZA[0] if not system.object.finite else ZA[2]
    a2 = a if np.ndim(a) else a*np.array(((-1., -1.), (1., 1.)))
    g = ra.GeometricTrace(system)
    g.rays_fields(FIELDS, yp, z, a)
    assert g.nrays == len(FIELDS)*len(yp)
    y0, u0 = np.asarray(g.y[0]), np.asarray(g.u[0])
    for f, yo in enumerate(FIELDS):
        with np.errstate(all="ignore"):
            yr, ur = oracle_rays(system, yo, yp, z, a2)
        sl = slice(f*len(yp), (f + 1)*len(yp))
        assert_parity(y0[sl][None], yr[None], 1e-13, key + ".y")
        assert_parity(u0[sl][None], ur[None], 1e-13, key + ".u")
    assert np.array_equal(np.asarray(g.i[0]), u0, equal_nan=True)
    assert not np.asarray(g.t[0]).any()
    # and the trace that follows equals the oracle's on the generated rays
    g.propagate(clip=True)
    table, ns = pack_system(system, g.l, g.n[0])
    with np.errstate(all="ignore"):
        want = tn.propagate(table, y0, u0, clip=True)
    for rows, b in zip((g.y, g.u, g.i, g.t), want):
        assert_parity(np.asarray(rows[1:]), b, RTOL_SPHERICAL, key)
    y1 = np.asarray(g.y[1])[:, :2]
    want_rms = np.sqrt((np.square(y1 - y1.mean(0)).sum(1)/g.nrays).sum())
    got_rms = g.rms(i=1)
    # the reference's equidistant projection yields a zero direction on axis
    # (NaN rays); NaN in, NaN out on both sides
    assert (np.isnan(got_rms) and np.isnan(want_rms)) or \
        got_rms == pytest.approx(want_rms, rel=1e-12)


@pytest.mark.gpu
def test_generation_fused_into_the_first_trace():
    """rt_generate_rays defers row 0 to the first trace, which builds the
    rays in registers: identical to the stand-alone generation + trace, for
    every way the batch can be touched first (trace, download of row 0,
    reduction, partial trace, non-default kernel variant, wavelength groups)."""
    from rayopt_amd.prescriptions import COOKE
    system = ra.system_from_yaml(COOKE % dict(
        air=1.0, sk16="1.62041/60.32", f2="1.62004/36.37"))
    fields = np.c_[np.zeros(5), np.linspace(0, 1, 5)]
    ref, yp, w = ra.pupil.pupil_distribution("hexapolar", 3000)
    from rayopt_amd.aiming import FieldAimer
    z, a = FieldAimer(system).pupil(fields)

    def run(first, fuse, **opts):
        g = ra.GeometricTrace(system)
        g.engine.set_option("fuse_generate", fuse)
        for k, v in opts.items():
            g.engine.set_option(k, v)
        g.rays_fields(fields, yp, z, a)
        out = {}
        if first == "row0":
            out["y0"] = np.array(np.asarray(g.y[0]))
        elif first == "rms0":
            out["r0"] = g.rms(i=0)
        elif first == "partial":
            g.propagate(stop=4, clip=True)
        g.propagate(clip=True)
        for name in "yuit":
            out[name] = np.array(np.asarray(getattr(g, name)))
        out["ms"] = g.kernel_ms()
        return out

    want = run("trace", 0)
    for first in ("trace", "row0", "rms0", "partial"):
        for opts in ({}, {"compact": 2}, {"alias_i": 0}):
            got = run(first, 1, **opts)
            for name in "yuit":
                assert np.array_equal(got[name], want[name], equal_nan=True), \
                    (first, opts, name)
    assert np.isfinite(want["y"][0]).all() and (want["t"][0] == 0).all()
    # several wavelengths in one generated batch
    g1, g0 = ra.GeometricTrace(system), ra.GeometricTrace(system)
    g0.engine.set_option("fuse_generate", 0)
    for g in (g1, g0):
        g.rays_points(fields, wavelength=system.wavelengths, nrays=300,
                      distribution="hexapolar", clip=True)
    for name in "yuit":
        assert np.array_equal(np.asarray(getattr(g1, name)),
                              np.asarray(getattr(g0, name)), equal_nan=True)


@pytest.mark.gpu
def test_retrace_of_a_generated_batch_rebuilds_the_rays():
    """A second propagate() of a device-generated batch builds the launch
    rays again in registers instead of reading row 0 (rt_trace: `regen`):
    same rows bit for bit as with the option off, row 0 untouched, also after
    the system changed between the traces; once the caller replaces a launch
    row, that row is what is traced."""
    from rayopt_amd.prescriptions import COOKE
    from rayopt_amd.aiming import FieldAimer
    from rayopt_amd._lib import RT_Y
    text = COOKE % dict(air=1.0, sk16="1.62041/60.32", f2="1.62004/36.37")
    fields = np.c_[np.zeros(4), np.linspace(0, 1, 4)]
    ref, yp, w = ra.pupil.pupil_distribution("hexapolar", 2500)

    def run(regen):
        system = ra.system_from_yaml(text)
        z, a = FieldAimer(system).pupil(fields)
        g = ra.GeometricTrace(system)
        g.engine.set_option("regenerate", regen)
        g.rays_fields(fields, yp, z, a)
        g.propagate(clip=True)
        first = {k: np.array(np.asarray(getattr(g, k))) for k in "yuit"}
        g.propagate(clip=True)              # the re-trace
        again = {k: np.array(np.asarray(getattr(g, k))) for k in "yuit"}
        g.propagate(stop=4, clip=True)      # ... of the first elements only
        g.propagate(start=4, clip=True)
        halves = {k: np.array(np.asarray(getattr(g, k))) for k in "yuit"}
        for k in "yuit":
            assert np.array_equal(halves[k][4:], again[k][4:], equal_nan=True)
            assert np.array_equal(halves[k][:1], again[k][:1], equal_nan=True)
        system[3].curvature *= 1.01         # another system, same rays
        system[0].curvature = 1e-3          # not what the rays were built on
        g.propagate(clip=False)
        moved = {k: np.array(np.asarray(getattr(g, k))) for k in "yuit"}
        # the caller's own launch heights from here on
        y0 = np.ascontiguousarray(first["y"][0].T)*.5
        g.engine.upload_row(RT_Y, 0, y0)
        for rows in (g.y, g.u, g.i, g.t):
            rows.invalidate(0, g.length)
        g.propagate(clip=False)
        halved = {k: np.array(np.asarray(getattr(g, k))) for k in "yuit"}
        return first, again, moved, halved

    on, off = run(1), run(0)
    for a, b in zip(on, off):
        for k in "yuit":
            assert np.array_equal(a[k], b[k], equal_nan=True), k
    first, again, moved, halved = on
    for k in "yuit":
        assert np.array_equal(first[k], again[k], equal_nan=True)
        assert np.array_equal(first[k][0], moved[k][0], equal_nan=True)
    assert not np.array_equal(first["y"][-1], moved["y"][-1], equal_nan=True)
    assert np.array_equal(halved["y"][0], first["y"][0]*.5)
    assert not np.array_equal(halved["y"][-1], moved["y"][-1], equal_nan=True)
