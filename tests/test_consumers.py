"""rms / refocus / opd: the oracle (oracle/consumers_numpy.py) against the
reference's own outputs, and the device-side reductions against both."""
import numpy as np
import pytest

import rayopt_amd as ra
from rayopt_amd.pack import pack_system
from oracle import trace_numpy as tn
from oracle import consumers_numpy as cn

from conftest import (consumer_golden_names, load_consumer_golden,
                      assert_parity)


def make_system(g):
    system = ra.system_from_yaml(g["yaml"])
    if g["finite_object"]:
        system.object = ra.Conjugate({"type": "finite", "radius": 1.}, True)
    return system


def oracle_arrays(system, g):
    n0 = system.refractive_index(g["l"], 0)
    table, ns = pack_system(system, g["l"], n0)
    Y, U, I, T = tn.propagate(table, g["y0"], g["u0"], clip=g["clip"])
    L, N = len(system), g["y0"].shape[0]
    full = lambda a, first: np.concatenate([first[None], a])   # noqa: E731
    return (full(Y, g["y0"]), full(U, g["u0"]), full(I, g["u0"]),
            full(T, np.zeros(N)), ns)


def close(a, b, rtol):
    if np.isnan(b):
        return np.isnan(a)
    return abs(a - b) <= rtol*abs(b)


@pytest.mark.parametrize("name", consumer_golden_names())
def test_oracle_consumers_match_reference(name):
    g = load_consumer_golden(name)
    system = make_system(g)
    Y, U, I, T, ns = oracle_arrays(system, g)
    with np.errstate(all="ignore"):
        assert close(cn.rms(Y[-1], g["w"]), g["rms_mean"], 1e-14)
        assert close(cn.rms(Y[-1], g["w"], g["ref"]), g["rms_ref"], 1e-14)
        assert close(cn.rms(Y[2], g["w"]), g["rms_mid"], 1e-14)
        w = g["w"] if g["w"] is not None else np.ones(len(g["y0"]))/len(g["y0"])
        assert close(cn.refocus_shift(Y[-1], I[-1], w), g["refocus_shift"],
                     1e-10)
        frames = [e.rot_normal if e.rotated else None for e in system]
        x, y, t = cn.opd_rays(Y, U, T, ns, g["ref"], system.origins, frames,
                              system.object.finite, g["radius"],
                              g["l"]/system.scale)
    assert_parity(x[None], g["opd_x"][None], 1e-12, name + ".x")
    assert_parity(y[None], g["opd_y"][None], 1e-12, name + ".y")
    assert_parity(t[None], g["opd_t"][None], 1e-9, name + ".t")


@pytest.mark.gpu
@pytest.mark.parametrize("name", consumer_golden_names())
def test_device_consumers_match_reference(name):
    """rt_rms / rt_refocus_shift / rt_opd_rays against the reference's
    numbers.  Tolerances: reductions 1e-12 (different summation order), the
    refocus ratio 1e-9 (centred sums cancel), OPD in waves 1e-9 (a ~70 mm
    path difference expressed in units of 0.6 um: condition ~1e5)."""
    g = load_consumer_golden(name)
    system = make_system(g)
    tr = ra.GeometricTrace(system)
    tr.rays_given(g["y0"], g["u0"], g["l"], g["w"], g["ref"])
    tr.propagate(clip=g["clip"])
    assert close(tr.rms(), g["rms_mean"], 1e-12)
    assert close(tr.rms(ref=g["ref"]), g["rms_ref"], 1e-12)
    assert close(tr.rms(i=2), g["rms_mid"], 1e-12)
    x, y, t = tr.opd(radius=g["radius"], resample=0)
    assert_parity(x[None], g["opd_x"][None], 1e-10, name + ".x")
    assert_parity(y[None], g["opd_y"][None], 1e-10, name + ".y")
    assert_parity(t[None], g["opd_t"][None], 1e-9, name + ".t")
    d0 = float(system[-1].distance)
    shift = tr.refocus()
    assert close(shift, g["refocus_shift"], 1e-9)
    assert float(system[-1].distance) == pytest.approx(d0 + shift, rel=1e-15)
    # refocus re-traced with the reference's defaults (clip=False)
    assert_parity(np.asarray(tr.y[-1])[None], g["y_after_refocus"][None],
                  1e-8, name + ".refocused")


@pytest.mark.gpu
def test_device_consumers_large_n_vs_oracle():
    """10^6 rays: device reductions against the numpy oracle on the arrays
    the device itself produced (isolates the reductions)."""
    system = ra.system_from_yaml(ra.prescriptions.DOUBLE_GAUSS)
    y, u = ra.bundles.disc_bundle(10**6, 17., 14., 3,
                                  ra.prescriptions.DOUBLE_GAUSS_PUPIL_Z)
    rng = np.random.default_rng(5)
    w = rng.random(10**6)
    w /= w.sum()
    tr = ra.GeometricTrace(system)
    tr.rays_given(y, u, None, w, 11)
    tr.propagate(clip=True)
    Y, I = np.asarray(tr.y[-1]), np.asarray(tr.i[-1])
    with np.errstate(all="ignore"):
        good = np.isfinite(Y[:, 0])
        assert 0.9 < good.mean() < 1.
        assert np.isnan(tr.rms())                       # NaN rays poison it
        # y[1] is finite for every ray (the intercept precedes the clip)
        assert np.isfinite(tr.rms(i=1))
        assert close(tr.rms(i=1), cn.rms(np.asarray(tr.y[1]), w), 1e-12)
        assert close(tr.rms(i=1, ref=11),
                     cn.rms(np.asarray(tr.y[1]), w, 11), 1e-12)
        assert close(tr.rms(i=3), cn.rms(np.asarray(tr.y[3]), w), 1e-12)
        assert tr.engine.refocus_shift(12) == pytest.approx(
            cn.refocus_shift(Y, I, w), rel=1e-9)
    x, yy, t = tr.opd(radius=100., resample=0)
    frames = [None]*len(system)
    xo, yo, to = cn.opd_rays(np.asarray(tr.y), np.asarray(tr.u),
                             np.asarray(tr.t), tr.n, 11, system.origins,
                             frames, False, 100., tr.l/system.scale)
    assert_parity(x[None], xo[None], 1e-12, "x")
    assert_parity(t[None], to[None], 1e-9, "t")
    xs, ys, ts = tr.opd(radius=100., resample=1)        # griddata path runs
    assert ts.shape == (1000, 1000) and np.isfinite(ts).any()


@pytest.mark.gpu
def test_psf_runs_and_is_normalised():
    system = ra.system_from_yaml(ra.prescriptions.SINGLET)
    y, u = ra.bundles.disc_bundle(2000, 4., 0., 1)
    tr = ra.GeometricTrace(system)
    tr.rays_given(y, u)
    tr.propagate()
    tr.refocus()
    p, q, psf = tr.psf(pad=2, resample=2)
    assert psf.shape == p.shape == q.shape
    assert np.isfinite(psf).all() and psf.max() > 0
    assert psf.sum() == pytest.approx(1., rel=0.2)


@pytest.mark.gpu
def test_resize_and_print_trace():
    system = ra.system_from_yaml(ra.prescriptions.DOUBLE_GAUSS)
    y, u = ra.bundles.disc_bundle(50000, 12., 5., 2,
                                  ra.prescriptions.DOUBLE_GAUSS_PUPIL_Z)
    tr = ra.GeometricTrace(system)
    tr.rays_given(y, u)
    tr.propagate()
    want = [np.hypot(*np.asarray(tr.y[j])[:, :2].T).max()
            for j in range(1, len(system))]
    tr.resize()
    got = [e.radius for e in system[1:]]
    np.testing.assert_allclose(got, want, rtol=1e-15)
    tr.resize(lambda a, b: max(a, b)*1.05)
    assert system[3].radius == pytest.approx(want[2]*1.05)
    text = "\n".join(tr.print_trace(rays=range(3)))
    assert text.count("ray ") == 3 and "height x" in text
    lines = list(tr.print_trace(rays=[7]))
    assert len(lines) == 2 + len(system) + 1
    # clipped rays make the maximum NaN, as np.max does in the reference
    system[5].radius = 1.0
    tr.propagate(clip=True)
    tr.resize()
    assert np.isnan(system[8].radius)


# -- per-bundle spot statistics (rt_spot_stats) -------------------------------

def test_oracle_spot_stats_is_reference_rms_per_bundle():
    """The grouped oracle equals the reference formula (``rms`` above, pinned
    to the reference by the goldens) applied to every bundle on its own."""
    rng = np.random.default_rng(3)
    P, G = 37, 5
    y = rng.normal(size=(P*G, 3))*[1., 2., 0.] + [3., -1., 0.]
    w = rng.random(P*G)
    for g in range(G):          # the reference normalises w per trace
        w[g*P:(g + 1)*P] /= w[g*P:(g + 1)*P].sum()
    s = cn.spot_stats(y, P, w)
    for g in range(G):
        sl = slice(g*P, (g + 1)*P)
        assert s[g, 0] == P and s[g, 5] == pytest.approx(1., rel=1e-14)
        assert np.sqrt(s[g, 3]) == pytest.approx(cn.rms(y[sl], w[sl]),
                                                 rel=1e-13)
        assert np.sqrt(s[g, 4]) == pytest.approx(
            np.hypot(*(y[sl, :2] - y[sl, :2].mean(0)).T).max(), rel=1e-14)
    # lost rays are left out and counted; an empty bundle gives NaN
    y[P + 3, 0] = np.nan
    y[2*P:3*P, 1] = np.nan
    s = cn.spot_stats(y, P, None)
    assert s[1, 0] == P - 1 and s[2, 0] == 0 and np.isnan(s[2, 1:5]).all()
    keep = np.r_[P:P + 3, P + 4:2*P]
    assert np.sqrt(s[1, 3]) == pytest.approx(
        cn.rms(y[keep]), rel=1e-13)


@pytest.mark.gpu
@pytest.mark.parametrize("groups,per", [(1, 100_000), (5, 200_000),
                                        (1000, 7), (64, 4096), (4001, 129)])
def test_device_spot_stats_vs_oracle(groups, per):
    """rt_spot_stats against the numpy oracle on the row the device itself
    produced: vignetted rays, weights, bundle sizes that are not multiples of
    anything, more bundles than workgroups per bundle."""
    system = ra.system_from_yaml(ra.prescriptions.DOUBLE_GAUSS)
    n = groups*per
    y, u = ra.bundles.disc_bundle(n, 17.5, 12., 7,
                                  ra.prescriptions.DOUBLE_GAUSS_PUPIL_Z)
    rng = np.random.default_rng(groups)
    w = rng.random(n)
    tr = ra.GeometricTrace(system)
    for weights in (None, w/w.sum()):
        tr.rays_given(y, u, None, weights, 0)
        tr.propagate(clip=True, keep=[-1])
        got = tr.spot_stats(group_rays=per)
        want = cn.spot_stats(np.asarray(tr.y[-1]), per, weights)
        assert got.shape == (groups, 6)
        assert np.array_equal(got[:, 0], want[:, 0])         # counts exact
        assert 0 < (want[:, 0] < per).sum()                  # some vignetted
        # the spread about the centroid inherits the centroid's summation
        # rounding magnified by |centroid| / spot size (~1e3 here)
        for c, rtol in ((1, 1e-12), (2, 1e-12), (3, 1e-9), (4, 1e-9),
                        (5, 1e-12)):
            assert_parity(got[None, :, c], want[None, :, c], rtol,
                          "stats column %d" % c)
    # run-to-run identical (no atomics)
    assert np.array_equal(tr.spot_stats(group_rays=per), got, equal_nan=True)
    with pytest.raises(ra.EngineError):
        tr.engine.spot_stats(len(system) - 1, per + 1, groups)
    with pytest.raises(ra.EngineError):
        tr.engine.spot_stats(3, per, groups)      # row not stored (keep)


# -- the row's statistics in one pass (rt_row_stats) --------------------------

def test_oracle_row_stats_is_the_reference_formulas():
    """The per-bundle oracle equals the reference's ``rms`` (about the mean
    and about a ray) and ``resize`` maximum applied to every bundle on its
    own; both are pinned to the reference by the consumer goldens."""
    rng = np.random.default_rng(5)
    P, G = 41, 4
    y = rng.normal(size=(P*G, 3))*[.3, .5, 0.] + [30., -7., 0.]
    w = rng.random(P*G)
    for g in range(G):
        w[g*P:(g + 1)*P] /= w[g*P:(g + 1)*P].sum()
    s = cn.row_stats(y, P, w, ref=3)
    for g in range(G):
        sl = slice(g*P, (g + 1)*P)
        assert s[g, 0] == P
        assert np.sqrt(s[g, 4]) == pytest.approx(cn.rms(y[sl], w[sl]),
                                                 rel=1e-13)
        assert np.sqrt(s[g, 5]) == pytest.approx(cn.rms(y[sl], w[sl], 3),
                                                 rel=1e-13)
        assert s[g, 6] == np.square(y[sl, :2]).sum(1).max()
    for name in consumer_golden_names():
        g = load_consumer_golden(name)
        Y = oracle_arrays(make_system(g), g)[0]
        if np.isnan(g["rms_mean"]):
            continue
        s = cn.row_stats(Y[-1], len(Y[-1]), g["w"], int(g["ref"]))[0]
        assert close(np.sqrt(s[4]), g["rms_mean"], 1e-13)
        assert close(np.sqrt(s[5]), g["rms_ref"], 1e-13)


@pytest.mark.gpu
@pytest.mark.parametrize("name", consumer_golden_names())
def test_row_stats_match_reference(name):
    """rt_row_stats against the REFERENCE's rms() numbers of the consumer
    goldens (about the mean, about the reference ray, a middle row), and the
    radius resize() takes."""
    g = load_consumer_golden(name)
    system = make_system(g)
    tr = ra.GeometricTrace(system)
    tr.rays_given(g["y0"], g["u0"], g["l"], g["w"], g["ref"])
    tr.propagate(clip=g["clip"])
    n = len(g["y0"])
    s = tr.row_stats()[0]
    lost = s["count"] < n
    # (a bundle that lost rays: the reference's numbers are NaN, ours cover
    # the survivors -- compared with the oracle below)
    assert lost == bool(np.isnan(g["rms_mean"]))
    if not lost:
        assert close(np.sqrt(s["var_mean"]), g["rms_mean"], 1e-12)
        assert close(np.sqrt(s["var_ref"]), g["rms_ref"], 1e-12)
        assert close(np.sqrt(tr.row_stats(i=2)[0]["var_mean"]),
                     g["rms_mid"], 1e-12)
        assert np.sqrt(s["r2_max"]) == pytest.approx(
            tr.engine.row_rmax(len(system) - 1), rel=1e-15)
    want = cn.row_stats(np.asarray(tr.y[-1]), n, g["w"], int(g["ref"]))[0]
    got = np.array(s.tolist())
    assert got[0] == want[0]
    for c, rtol in ((1, 1e-12), (2, 1e-12), (3, 1e-12), (4, 1e-9), (5, 1e-9),
                    (6, 1e-15), (7, 1e-12), (8, 1e-12)):
        assert close(got[c], want[c], rtol), (name, c, got[c], want[c])


@pytest.mark.gpu
@pytest.mark.parametrize("groups,per,block", [
    (1, 100_000, 0), (5, 200_000, 0), (1000, 7, 0), (64, 4096, 0),
    (4001, 129, 0), (5, 60_001, 4096), (3, 1, 0), (4200, 65, 0)])
def test_device_row_stats_vs_oracle(groups, per, block):
    """rt_row_stats against the numpy oracle on the row the device itself
    produced: vignetted rays (also the reference ray's, also the bundle's
    first rays), weights, bundle sizes that are multiples of nothing, more
    bundles than fit the pinned result, a batch in blocks; and against the
    three calls it replaces."""
    system = ra.system_from_yaml(ra.prescriptions.DOUBLE_GAUSS)
    n = groups*per
    y, u = ra.bundles.disc_bundle(n, 17.5, 12., 7,
                                  ra.prescriptions.DOUBLE_GAUSS_PUPIL_Z)
    rng = np.random.default_rng(groups)
    w = rng.random(n) + .01
    eng = ra.Engine()
    if block:
        eng.set_option("block_rays", block)
    tr = ra.GeometricTrace(system, engine=eng)
    L = len(system)
    for weights, ref in ((None, -1), (w/w.sum(), min(per - 1, 3)),
                         (None, 0)):
        tr.rays_given(y, u, None, weights, max(ref, 0))
        tr.propagate(clip=True, keep=[-1])
        got = eng.row_stats(L - 1, per, groups, ref)
        Y = np.asarray(tr.y[-1])
        want = cn.row_stats(Y, per, weights, ref)
        assert got.shape == (groups, 10)
        assert np.array_equal(got[:, 0], want[:, 0])         # counts exact
        if per > 100:
            assert 0 < (want[:, 0] < per).sum()              # some vignetted
        for c, rtol in ((1, 1e-12), (2, 1e-12), (3, 1e-12), (4, 1e-9),
                        (5, 1e-9), (6, 1e-15), (7, 1e-12), (8, 1e-12)):
            assert_parity(got[None, :, c], want[None, :, c], rtol,
                          "row_stats column %d" % c)
        # the calls it replaces
        spot = eng.spot_stats(L - 1, per, groups)
        assert np.array_equal(got[:, 0], spot[:, 0])
        assert_parity(got[None, :, 4], spot[None, :, 3], 1e-9, "vs spot")
        rmax = eng.row_rmax(L - 1)
        if np.isfinite(rmax):
            assert np.sqrt(got[:, 6].max()) == rmax
        # run-to-run identical (no atomics in the sums)
        assert np.array_equal(eng.row_stats(L - 1, per, groups, ref), got,
                              equal_nan=True)
    with pytest.raises(ra.EngineError):
        eng.row_stats(L - 1, per + 1, groups)
    with pytest.raises(ra.EngineError):
        eng.row_stats(L - 1, per, groups, per)     # no such ray in a bundle
    with pytest.raises(ra.EngineError):
        eng.row_stats(3, per, groups)              # row not stored (keep)


@pytest.mark.gpu
def test_row_stats_far_from_the_axis_and_a_lost_shift_ray():
    """The shift is a ray of the bundle, so a spot far from the axis costs no
    bits (spot 1e-3 at 1e3 from the axis: about the mean to 1e-9 of itself);
    a bundle whose first 300 rays are lost still has its count and maximum,
    and its spread through the two-pass fallback."""
    system = ra.system_from_yaml(ra.prescriptions.SINGLET)
    n = 4096
    rng = np.random.default_rng(8)
    Y = np.zeros((n, 3))
    Y[:, :2] = [1e3, -2e3] + 1e-3*rng.standard_normal((n, 2))
    tr = ra.GeometricTrace(system)
    y, u = ra.bundles.disc_bundle(n, 5., 0., 0)
    tr.rays_given(y, u)
    tr.propagate()
    from rayopt_amd._lib import RT_Y
    L = len(system)
    for lost in (0, 300):
        row = Y.copy()
        row[:lost] = np.nan
        tr.engine.upload_row(RT_Y, L - 1, np.ascontiguousarray(row.T))
        got = tr.engine.row_stats(L - 1, n, 1, -1)[0]
        want = cn.row_stats(row, n, None, None)[0]
        assert got[0] == want[0] == n - lost
        for c, rtol in ((2, 1e-14), (3, 1e-14), (4, 1e-9), (6, 1e-15)):
            assert close(got[c], want[c], rtol), (lost, c, got[c], want[c])


@pytest.mark.gpu
def test_xy_of_a_row_cross_pcie_alone():
    """``t.y[-1, :, :2]`` -- what rms() and the spot diagrams of the
    reference read (rayopt/geometric_trace.py:172, analysis.py:237-283) --
    brings down x and y only (rt_download_xy); the full row, asked for
    later, is the full row."""
    system = ra.system_from_yaml(ra.prescriptions.DOUBLE_GAUSS)
    n = 70_001
    y, u = ra.bundles.disc_bundle(n, 17., 9., 5,
                                  ra.prescriptions.DOUBLE_GAUSS_PUPIL_Z)
    for block in (0, 4096):
        eng = ra.Engine()
        if block:
            eng.set_option("block_rays", block)
        a = ra.GeometricTrace(system, engine=eng)
        b = ra.GeometricTrace(system)
        for t in (a, b):
            t.rays_given(y, u)
            t.propagate(clip=True)
        full = np.asarray(b.y[-1])
        xy = a.y[-1, :, :2]
        assert xy.shape == (n, 2) and not a.y._valid[-1]
        assert np.array_equal(xy, full[:, :2], equal_nan=True)
        assert np.array_equal(a.u[-1, :, 1], np.asarray(b.u[-1])[:, 1],
                              equal_nan=True)
        assert np.array_equal(a.i[5, :, 0:1], np.asarray(b.i[5])[:, 0:1],
                              equal_nan=True)
        assert np.array_equal(a.y[-1], full, equal_nan=True)   # now all of it
        assert np.array_equal(a.y[-1, :, 2], full[:, 2], equal_nan=True)
        # a new trace voids both
        a.propagate(clip=False)
        assert not a.y._valid_xy[-1] and not a.y._valid[-1]
        assert np.array_equal(a.y[-1, :, :2],
                              np.asarray(a.y[-1])[:, :2], equal_nan=True)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 2, 3, 63, 64, 257, 1001, 99_999, 1_000_002])
def test_one_pass_reductions_vs_two_pass_and_numpy(n):
    """rt_rms / rt_refocus_shift / rt_row_rmax sum in ONE pass over the rows
    (shifted by ray 0, 16-byte loads: csrc/rt_consumer_kernels.h).  Against
    the two-pass kernels ("consumers_one_pass" = 0) and against numpy on the
    rows the device produced, for ray counts around every boundary of the
    load pattern (odd, one pair, one wavefront, beyond the grid)."""
    system = ra.system_from_yaml(ra.prescriptions.DOUBLE_GAUSS)
    y, u = ra.bundles.disc_bundle(n, 12., 9., 21,
                                  ra.prescriptions.DOUBLE_GAUSS_PUPIL_Z)
    rng = np.random.default_rng(n)
    w = rng.random(n) + .1
    tr = ra.GeometricTrace(system)
    L = len(system)
    for weights in (None, w/w.sum(), w):                # w need not sum to 1
        tr.rays_given(y, u, None, weights, n//2)
        tr.propagate(clip=False)
        Y, I = np.asarray(tr.y[-1]), np.asarray(tr.i[-1])
        assert np.isfinite(Y).all()
        got = {}
        for one in (1, 0):
            tr.engine.set_option("consumers_one_pass", one)
            got[one] = (tr.rms(), tr.rms(ref=n//2), tr.rms(i=3),
                        tr.engine.refocus_shift(L - 1) if n > 2 else 0.,
                        tr.engine.row_rmax(L - 1))
        tr.engine.set_option("consumers_one_pass", 1)
        want = (cn.rms(Y, weights), cn.rms(Y, weights, n//2),
                cn.rms(np.asarray(tr.y[3]), weights),
                cn.refocus_shift(Y, I, weights) if n > 2 else 0.,
                np.sqrt(np.square(Y[:, :2]).sum(1).max()))
        for k, rtol in enumerate((1e-12, 1e-12, 1e-12, 1e-9, 1e-15)):
            assert got[1][k] == pytest.approx(want[k], rel=rtol, abs=1e-300), k
            assert got[0][k] == pytest.approx(want[k], rel=rtol, abs=1e-300), k
        assert got[1] == (tr.rms(), tr.rms(ref=n//2), tr.rms(i=3),
                          tr.engine.refocus_shift(L - 1) if n > 2 else 0.,
                          tr.engine.row_rmax(L - 1))     # run-to-run identical


@pytest.mark.gpu
def test_one_pass_reductions_fall_back_when_the_shift_is_poor():
    """Ray 0 far outside the bundle: the shifted sums would cancel (|mean -
    ray 0| >> spot size); the library notices (what the subtraction started
    from is reported by the kernel) and runs the two passes.  Same tolerance
    as ever.  Ray 0 vignetted: refocus (which leaves such rays out) still
    answers; rms is NaN as in the reference."""
    system = ra.system_from_yaml(ra.prescriptions.DOUBLE_GAUSS)
    n = 50_000
    y, u = ra.bundles.disc_bundle(n, 2., 9., 5,
                                  ra.prescriptions.DOUBLE_GAUSS_PUPIL_Z)
    tr = ra.GeometricTrace(system)
    L = len(system)
    # a tight bundle and one ray of another field
    u2 = u.copy()
    u2[0, 1] += .02
    u2[0] /= np.sqrt(np.square(u2[0]).sum())
    tr.rays_given(y, u2, None, None, 1)
    tr.propagate(clip=False)
    Y, I = np.asarray(tr.y[-1]), np.asarray(tr.i[-1])
    d = Y[:, :2] - Y[1:, :2].mean(0)
    assert np.hypot(*d[0]) > 30*np.sqrt(np.square(d[1:]).sum(1).mean())
    # the poor shift as numpy sees it: |m|^2 >> variance
    assert tr.rms() == pytest.approx(cn.rms(Y, None), rel=1e-12)
    assert tr.engine.refocus_shift(L - 1) == pytest.approx(
        cn.refocus_shift(Y, I, None), rel=1e-9)
    # the same rows with a far-away origin: every ray is "far from zero" but
    # close to ray 0 -- the case the shift is there for
    tr.rays_given(y, u, None, None, 1)
    tr.propagate(clip=False)
    Y = np.asarray(tr.y[-1])
    assert tr.rms() == pytest.approx(cn.rms(Y, None), rel=1e-12)
    # ray 0 vignetted
    y3 = y.copy()
    y3[0, :2] = 500.
    tr.rays_given(y3, u, None, None, 1)
    tr.propagate(clip=True)
    Y, I = np.asarray(tr.y[-1]), np.asarray(tr.i[-1])
    assert np.isnan(Y[0, 0]) and np.isfinite(Y[1:]).all()
    assert np.isnan(tr.rms())
    assert tr.engine.refocus_shift(L - 1) == pytest.approx(
        cn.refocus_shift(Y, I, None), rel=1e-9)


@pytest.mark.gpu
def test_reductions_repeat_bit_for_bit():
    """500 calls of each reduction on 3*10^6 rays, two contexts on the same
    device interleaved (their partials and pinned scalars must not be
    shared): every answer the same bits."""
    system = ra.system_from_yaml(ra.prescriptions.DOUBLE_GAUSS)
    n = 3_000_000
    y, u = ra.bundles.disc_bundle(n, 12., 5., 1,
                                  ra.prescriptions.DOUBLE_GAUSS_PUPIL_Z)
    a, b = ra.GeometricTrace(system), ra.GeometricTrace(system)
    L = len(system)
    for tr, m in ((a, n), (b, n//3)):
        tr.rays_given(y[:m], u[:m])
        tr.propagate(clip=False)
    first = {}
    for rep in range(500):
        for tr in (a, b):
            got = (tr.rms(), tr.engine.refocus_shift(L - 1),
                   tr.engine.row_rmax(L - 1), tr.rms(ref=5))
            assert np.isfinite(got).all()
            assert first.setdefault(id(tr), got) == got, rep
    Y = np.asarray(a.y[-1])
    assert first[id(a)][0] == pytest.approx(cn.rms(Y, None), rel=1e-12)
    assert first[id(a)][2] == pytest.approx(
        np.sqrt(np.square(Y[:, :2]).sum(1).max()), rel=1e-15)


def opd_stats_numpy(x, y, t, w=None):
    """What rt_opd_stats promises, from per-ray values: over the rays where
    x, y, t are finite (the rays the reference keeps,
    rayopt/geometric_trace.py:133-135)."""
    good = np.isfinite(x) & np.isfinite(y) & np.isfinite(t)
    w = np.ones(len(t)) if w is None else np.asarray(w)
    wg, tg = w[good], t[good]
    if not good.any():
        return np.r_[0., 0., [np.nan]*6]
    mean = (wg*tg).sum()/wg.sum()
    return np.array([good.sum(), wg.sum(), mean,
                     np.sqrt((wg*(tg - mean)**2).sum()/wg.sum()),
                     tg.min(), tg.max(), tg.max() - tg.min(),
                     np.sqrt((wg*tg**2).sum()/wg.sum())])


def assert_opd_stats(got, want, name):
    assert got[0] == want[0], name
    if want[0] == 0:
        assert np.isnan(got[2:]).all(), name
        return
    scale = max(abs(want[4]), abs(want[5]), 1e-300)
    # sums of 1e-9-accurate path differences (waves): absolute against the
    # size of the map
    np.testing.assert_allclose(got[1], want[1], rtol=1e-12, err_msg=name)
    np.testing.assert_allclose(got[2:], want[2:], rtol=0, atol=1e-9*scale,
                               err_msg=name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", consumer_golden_names())
def test_opd_stats_match_reference(name):
    """rt_opd_stats (mean / rms / P-V of the OPD, reduced on the device, no
    per-ray copy) against the statistics of the REFERENCE's per-ray opd()
    values (rayopt/geometric_trace.py:101-131, resample=0), weights and
    vignetted rays included; the x | y | t array it can leave on the device
    is the array rt_opd_rays returns, bit for bit."""
    g = load_consumer_golden(name)
    system = make_system(g)
    tr = ra.GeometricTrace(system)
    tr.rays_given(g["y0"], g["u0"], g["l"], g["w"], g["ref"])
    tr.propagate(clip=g["clip"])
    with np.errstate(all="ignore"):
        want = opd_stats_numpy(g["opd_x"], g["opd_y"], g["opd_t"], g["w"])
    got = tr.opd_stats(radius=g["radius"], keep=True)
    assert got.shape == (1, 8)
    assert_opd_stats(got[0], want, name)
    ptr, n = tr.engine.opd_device()
    assert n == tr.nrays
    kept = tr.engine.copy_to_host(ptr, 3*n*8).reshape(3, n)
    x, y, t = tr.opd(radius=g["radius"], resample=0)
    assert np.array_equal(kept, np.array([x, y, t]), equal_nan=True)
    # a new trace of the same rays voids what was kept: the array described
    # the rows as they were
    tr.opd_stats(radius=g["radius"], keep=True)
    tr.propagate(clip=g["clip"])
    with pytest.raises(ra.EngineError, match="no path differences"):
        tr.engine.opd_device()
    # without keep nothing is left behind for this batch ...
    tr.rays_given(g["y0"], g["u0"], g["l"], g["w"], g["ref"])
    tr.propagate(clip=g["clip"])
    assert_opd_stats(tr.opd_stats(radius=g["radius"])[0], want, name)
    with pytest.raises(ra.EngineError, match="no path differences"):
        tr.engine.opd_device()


@pytest.mark.gpu
@pytest.mark.parametrize("n,bundles,block", [(60_000, 5, 0), (64*300, 3, 4096),
                                             (257*7, 7, 512), (1000, 1, 256)])
def test_opd_stats_per_bundle(n, bundles, block):
    """Several field bundles in one batch, each against its own reference
    ray, in the plain layout and in blocks (bundles straddle blocks); the
    per-bundle numbers are those of the bundle traced on its own."""
    system = ra.system_from_yaml(ra.prescriptions.DOUBLE_GAUSS)
    per = n//bundles
    rng = np.random.default_rng(n)
    ys, us = [], []
    for k in range(bundles):
        y, u = ra.bundles.disc_bundle(per, 15., 12.*k/max(bundles - 1, 1), k,
                                      ra.prescriptions.DOUBLE_GAUSS_PUPIL_Z)
        ys.append(y)
        us.append(u)
    y, u = np.concatenate(ys), np.concatenate(us)
    w = rng.random(n) + .1
    ref = per//3
    for weights in (None, w):
        tr = ra.GeometricTrace(system)
        if block:
            tr.engine.set_option("block_rays", block)
        tr.rays_given(y, u, None, weights, ref)
        tr.propagate(clip=True)
        assert (tr.engine.blocks()[0] > 1) == bool(block)
        got = tr.opd_stats(radius=100., bundles=bundles)
        assert got.shape == (bundles, 8)
        for k in range(bundles):
            one = ra.GeometricTrace(system)
            sl = slice(k*per, (k + 1)*per)
            one.rays_given(y[sl], u[sl], None,
                           None if weights is None else weights[sl], ref)
            one.propagate(clip=True)
            x1, y1, t1 = one.opd(radius=100., resample=0)
            with np.errstate(all="ignore"):
                want = opd_stats_numpy(x1, y1, t1,
                                       None if weights is None
                                       else weights[sl])
            assert 0 < want[0] <= per
            assert_opd_stats(got[k], want, "bundle %d" % k)
    with pytest.raises(ValueError, match="do not split"):
        tr.opd_stats(radius=100.,
                     bundles=next(k for k in range(2, n) if n % k))


@pytest.mark.gpu
def test_opd_rays_takes_its_host_buffer_again_only_when_nobody_holds_it():
    """The per-ray OPD lands in the buffer of the previous call (fresh host
    pages cost more than PCIe) -- unless the caller still holds that result
    or a view of it, which then stays what it was."""
    system = ra.system_from_yaml(ra.prescriptions.DOUBLE_GAUSS)
    y, u = ra.bundles.disc_bundle(50_000, 12., 3., 1,
                                  ra.prescriptions.DOUBLE_GAUSS_PUPIL_Z)
    tr = ra.GeometricTrace(system)
    tr.rays_given(y, u)
    tr.propagate()
    x1, y1, t1 = tr.opd_rays(radius=100.)
    keep = t1.copy()
    y2, u2 = ra.bundles.disc_bundle(50_000, 12., 9., 2,
                                    ra.prescriptions.DOUBLE_GAUSS_PUPIL_Z)
    tr.rays_given(y2, u2)
    tr.propagate()
    x2, yy2, t2 = tr.opd_rays(radius=100.)     # t1 is still held
    assert np.array_equal(t1, keep, equal_nan=True)
    assert not np.shares_memory(t1, t2)
    assert not np.array_equal(t1, t2, equal_nan=True)
    where = x2.ctypes.data
    del x2, yy2, t2
    x3, y3, t3 = tr.opd_rays(radius=100.)      # nobody held the second
    assert x3.ctypes.data == where
