"""Parity of the HIP engine with the reference, on a real MI355X.

Every test goes through the C ABI (librt_mi355.so via ctypes) and compares
with (a) the committed golden vectors produced by the unmodified reference
and (b) the numpy oracle run on the same seeded inputs.  Tolerances are the
contract's: 1e-10 relative for plane/sphere/conic surfaces, 1e-8 for iterated
aspheres, identical NaN masks for y, u, i, t separately.
"""
import numpy as np
import pytest

import rayopt_amd as ra
from rayopt_amd import prescriptions as P
from rayopt_amd.bundles import disc_bundle, multi_field_bundle
from rayopt_amd.pack import pack_system, resolve_range
from oracle import trace_numpy as tn

from conftest import (golden_names, load_golden, assert_parity, case_rtol,
                      RTOL_SPHERICAL, RTOL_ASPHERE, same_or_contract)

pytestmark = pytest.mark.gpu


def oracle_trace(system, y, u, l, clip, start=1, stop=None):
    a, b = resolve_range(len(system), start, stop)
    table, ns = pack_system(system, l, system.refractive_index(l, 0), a, b)
    return tn.propagate(table, y, u, a, b, clip), ns


def gpu_trace(system, y, u, l, clip, start=1, stop=None, **options):
    g = ra.GeometricTrace(system)
    for k, v in options.items():
        g.engine.set_option(k, v)
    g.rays_given(y, u, l)
    g.propagate(start=start, stop=stop, clip=clip)
    return g


def compare(g, want, a, b, rtol, what):
    for label, rows, ref in (("y", g.y, want[0]), ("u", g.u, want[1]),
                             ("i", g.i, want[2]), ("t", g.t, want[3])):
        assert_parity(np.asarray(rows[a:b]), ref, rtol,
                      "%s.%s" % (what, label))


@pytest.mark.parametrize("name", golden_names())
def test_matches_reference_golden(name, arith):
    gold = load_golden(name)
    aspheric = "aspherics" in gold["yaml"]
    if arith == "default" and not aspheric:
        pytest.skip("no aspheric element: one arithmetic")
    system = ra.system_from_yaml(gold["yaml"])
    a, b = resolve_range(len(system), gold["start"], gold["stop"])
    g = gpu_trace(system, gold["y0"], gold["u0"], gold["l"], gold["clip"],
                  gold["start"], gold["stop"])
    want = [gold[k][a:b] for k in "yuit"]
    compare(g, want, a, b, case_rtol(gold), name)
    # ... and to the last bit: plane / sphere / conic, tilted or not (the 3x3
    # products follow the dgemm's fused chain), iterated aspheres (scipy's
    # Newton operation for operation, BLAS-summed derivative included)
    # (the default arithmetic for aspheres: contract + identical NaN masks)
    for rows, ref, label in zip((g.y, g.u, g.i, g.t), want, "yuit"):
        same_or_contract(rows[a:b], ref, aspheric, arith, (name, label))
    assert np.array_equal(g.n[:b], gold["n"][:b])
    # row 0 is what rays_given stored
    assert np.array_equal(g.y[0], gold["y0"])
    assert np.array_equal(g.u[0], gold["u0"])
    assert np.array_equal(g.i[0], gold["u0"])
    assert not np.asarray(g.t[0]).any()


@pytest.mark.parametrize("options", [
    dict(alias_i=0), dict(compact=2), dict(compact=2, compact_every=1),
    dict(regenerate=0, fuse_generate=0), dict(resident_lds=0),
    dict(resident_lds=65536), dict(resident_lds=20480),
    dict(uniform_input=0), dict(range_shortcuts=0),
    dict(range_shortcuts=0, resident_lds=32768)])
@pytest.mark.parametrize("key", ["double_gauss", "asphere_phone", "torture"])
def test_kernel_variants_are_bit_identical(key, options):
    system = ra.system_from_yaml(P.ALL[key])
    rad = min(float(e.radius) for e in system[1:-1])*1.1
    y, u = disc_bundle(10007, rad, 2., 21)     # ragged: not a multiple of 64
    base = gpu_trace(system, y, u, None, True)
    var = gpu_trace(system, y, u, None, True, **options)
    for a, b in ((base.y, var.y), (base.u, var.u), (base.i, var.i),
                 (base.t, var.t)):
        assert np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


def test_kat_reference_rms():
    """The reference's pinned known answer (test_raytrace.py:189-199)."""
    with np.load("tests/golden/kat_cooke_quadrature.npz") as z:
        k = {key: z[key] for key in z.files}
    system = ra.system_from_yaml(P.cooke())
    for el, n in zip(system, k["n"]):
        if el.material is not None:
            el.material = ra.ConstantIndex(float(n))
    g = ra.GeometricTrace(system)
    g.rays_given(k["y0"], k["u0"], float(k["l"]), k["w"], int(k["ref"]))
    g.propagate(clip=False)
    assert_parity(np.asarray(g.y), k["y"], RTOL_SPHERICAL, "kat.y")
    y = g.y[-1][:, :2]
    rms = np.sqrt((np.square(y - y.mean(0)).sum(1)*g.w).sum())
    assert rms == pytest.approx(float(k["rms"]), rel=1e-10)
    np.testing.assert_allclose(rms, .052, rtol=1e-2)


# ---- BASELINE.json configs against the oracle on the same seeded rays ----

def test_config_c1_singlet_1e4():
    system = ra.system_from_yaml(P.SINGLET)
    y, u = disc_bundle(10**4, 8.0, 0., 0)
    g = gpu_trace(system, y, u, None, True)
    want, ns = oracle_trace(system, y, u, g.l, True)
    compare(g, want, 1, 4, RTOL_SPHERICAL, "C1")
    assert np.isfinite(np.asarray(g.y)).all()


@pytest.mark.parametrize("l", [587.56e-9, 656.27e-9, 486.13e-9])
def test_config_c2_cooke_1e6(l):
    system = ra.system_from_yaml(P.cooke(l))
    y, u = disc_bundle(10**6, 5.5, 5., 0)
    g = gpu_trace(system, y, u, l, True)
    want, ns = oracle_trace(system, y, u, l, True)
    compare(g, want, 1, 9, RTOL_SPHERICAL, "C2")
    assert np.array_equal(g.n[1:], ns[1:])


def _c3_rays(n, seed=0):
    th = [f*P.DOUBLE_GAUSS_FIELD_DEG for f in (0, .35, .5, .7, 1.)]
    return multi_field_bundle(n, 17., th, seed, P.DOUBLE_GAUSS_PUPIL_Z)


def test_config_c3_double_gauss_1e6_vs_oracle():
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    y, u = _c3_rays(10**6)
    g = gpu_trace(system, y, u, None, True)
    want, ns = oracle_trace(system, y, u, g.l, True)
    compare(g, want, 1, 13, RTOL_SPHERICAL, "C3")
    frac = np.isnan(np.asarray(g.u[-1])[:, 0]).mean()
    assert 0.001 < frac < 0.1      # a few per cent vignetted, as designed


def test_config_c4_asphere_2e5_vs_oracle(arith):
    system = ra.system_from_yaml(P.ASPHERE_PHONE)
    ys, us = [], []
    for k, deg in enumerate((0., 17.5)):
        y, u = disc_bundle(10**5, 0.6, deg, k)
        y[:, 1] -= 0.5*np.tan(np.radians(deg))
        ys.append(y)
        us.append(u)
    y, u = np.concatenate(ys), np.concatenate(us)
    g = gpu_trace(system, y, u, None, True)
    want, ns = oracle_trace(system, y, u, g.l, True)
    compare(g, want, 1, 9, RTOL_ASPHERE, "C4")
    for rows, ref, label in zip((g.y, g.u, g.i, g.t), want, "yuit"):
        same_or_contract(rows[1:9], ref, True, arith, "C4." + label)
    assert np.isfinite(np.asarray(g.y[-1])).mean() > 0.99


# ---- full size (10^7 rays): size-independent properties -----------------

def _subsample_check(system, g, y, u, clip, rtol, step):
    """Every ``step``-th ray of the full-size device result against the
    oracle run on just those rays (rays are independent)."""
    sel = slice(0, None, step)
    want, ns = oracle_trace(system, y[sel], u[sel], g.l, clip)
    L = len(system)
    for label, rows, ref in (("y", g.y, want[0]), ("u", g.u, want[1]),
                             ("i", g.i, want[2]), ("t", g.t, want[3])):
        for j in range(1, L):
            assert_parity(np.asarray(rows[j])[sel][None], ref[j - 1][None],
                          rtol, "full.%s[%d]" % (label, j))


def test_full_size_c3_properties():
    n = 10**7
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    y, u = _c3_rays(n)
    g = gpu_trace(system, y, u, None, True)
    L = len(system)
    # (1) subsample against the oracle
    _subsample_check(system, g, y, u, True, RTOL_SPHERICAL, 97)
    # (2) shard invariance: tracing a contiguous shard alone gives the same
    # bits as the same rays inside the big batch (rays are independent)
    lo, hi = 3_000_001, 3_200_000
    part = gpu_trace(system, y[lo:hi], u[lo:hi], None, True)
    for rows, prow in ((g.y, part.y), (g.u, part.u), (g.i, part.i)):
        assert np.array_equal(np.asarray(rows[-1])[lo:hi],
                              np.asarray(prow[-1]), equal_nan=True)
    assert np.array_equal(np.asarray(g.t[5])[lo:hi], np.asarray(part.t[5]),
                          equal_nan=True)
    # (3) physics invariants on the full arrays of a few rows
    dead_before = np.zeros(n, dtype=bool)
    for j in range(1, L):
        uj = np.asarray(g.u[j])
        dead = np.isnan(uj[:, 0])
        assert not (dead_before & ~dead).any()      # NaN is absorbing
        norm = np.square(uj[~dead]).sum(1)
        assert np.abs(norm - 1).max() < 1e-13       # directions stay unit
        if j in (1, 6, L - 1):
            ij = np.asarray(g.i[j])
            prev = np.asarray(g.u[j - 1])
            # unrotated system: incoming direction == previous outgoing
            assert np.array_equal(ij, prev, equal_nan=True)
        dead_before = dead
    t = np.asarray(g.t[1:])
    assert (t[np.isfinite(t)] > -1e-9).all()        # forward propagation


def test_full_size_c4_asphere_subsample(arith):
    n = 10**7
    system = ra.system_from_yaml(P.ASPHERE_PHONE)
    y, u = disc_bundle(n, 0.6, 0.7*25, 3)
    y[:, 1] -= 0.5*np.tan(np.radians(0.7*25))
    g = gpu_trace(system, y, u, None, True)
    _subsample_check(system, g, y, u, True, RTOL_ASPHERE, 997)


# ---- API behaviour ---------------------------------------------------------

def test_partial_repropagate_like_refocus():
    """refocus() edits system[-1].distance and re-propagates
    (geometric_trace.py:98-99); re-running only the last element from the
    stored row L-2 must equal a full trace of the edited system."""
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    y, u = _c3_rays(20000)
    g = gpu_trace(system, y, u, None, True)
    system[-1].distance += 0.37
    full = gpu_trace(system, y, u, None, True)
    g.propagate(start=len(system) - 1, clip=True)
    for a, b in ((g.y, full.y), (g.u, full.u), (g.i, full.i), (g.t, full.t)):
        assert np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


def test_start_from_rotated_element():
    """start > 1 seeds from a row in a tilted element's frame: the kernel
    applies that element's from_normal first (geometric_trace.py:75-76)."""
    system = ra.system_from_yaml(P.TORTURE)
    y, u = disc_bundle(5000, 9., 2., 5)
    g = gpu_trace(system, y, u, None, True)
    want, _ = oracle_trace(system, y, u, g.l, True)
    before = [np.array(np.asarray(r)) for r in (g.y, g.u, g.i, g.t)]
    g.propagate(start=4, stop=8, clip=True)
    for rows, old in zip((g.y, g.u, g.i, g.t), before):
        assert np.array_equal(np.asarray(rows), old, equal_nan=True)
    compare(g, want, 1, 9, RTOL_SPHERICAL, "torture")


def test_system_propagate_generator():
    """System.propagate yields (y,u,n,i,t) per element (system.py:459-464)."""
    system = ra.system_from_yaml(P.TORTURE)
    y, u = disc_bundle(3000, 12., 3., 8)
    l = system.wavelengths[0]
    table, ns = pack_system(system, l, 1.0)
    Y, U, I, T = tn.propagate(table, y, u, clip=True)
    rows = list(system.propagate(y, u, 1.0, l, clip=True))
    assert len(rows) == len(system) - 1
    for j, (yj, uj, nj, ij, tj) in enumerate(rows):
        assert_parity(yj[None], Y[j][None], RTOL_SPHERICAL, "gen.y")
        assert_parity(uj[None], U[j][None], RTOL_SPHERICAL, "gen.u")
        assert_parity(ij[None], I[j][None], RTOL_SPHERICAL, "gen.i")
        assert_parity(tj[None], T[j][None], RTOL_SPHERICAL, "gen.t")
        assert nj == ns[j + 1]


def test_rays_given_two_components_and_broadcast():
    system = ra.system_from_yaml(P.SINGLET)
    y = np.array([[0., 1.], [2., -3.], [0.5, 0.25]])
    u = np.array([[0., .01]])                       # broadcast, u_z completed
    g = ra.GeometricTrace(system)
    g.rays_given(y, u)
    g.propagate()
    assert g.y.shape == (4, 3, 3) and g.t.shape == (4, 3)
    u0 = np.asarray(g.u[0])
    assert np.allclose(u0[:, 2], np.sqrt(1 - .01**2))
    y3 = np.zeros((3, 3))
    y3[:, :2] = y
    want, _ = oracle_trace(system, y3, u0, g.l, False)
    compare(g, want, 1, 4, RTOL_SPHERICAL, "2comp")
    assert np.allclose(g.w, 1/3) and g.ref == 0


def test_lazy_rows_views():
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    y, u = _c3_rays(1000)
    g = gpu_trace(system, y, u, None, True)
    full = np.asarray(g.y)
    assert full.shape == (13, 1000, 3)
    assert np.array_equal(g.y[-1], full[-1], equal_nan=True)
    assert np.array_equal(g.y[2:5], full[2:5], equal_nan=True)
    assert np.array_equal(g.y[-1, :, :2], full[-1, :, :2], equal_nan=True)
    assert np.array_equal(g.t[:-1].sum(0), np.asarray(g.t)[:-1].sum(0),
                          equal_nan=True)
    assert len(g.y) == 13 and g.t.ndim == 2


def test_errors_are_loud():
    system = ra.system_from_yaml(P.SINGLET)
    g = ra.GeometricTrace(system)
    with pytest.raises(ValueError, match="rays_given"):
        g.propagate()
    with pytest.raises(ValueError):
        g.rays_given(np.zeros((4, 1)), np.zeros((4, 1)))
    with pytest.raises(ValueError):
        g.rays_given(np.zeros((4, 3)), np.zeros((4, 3)), w=np.ones(3))
    g.rays_given(np.zeros((4, 3)), np.array([[0, 0, 1.]]))
    with pytest.raises(ValueError):
        g.propagate(start=0)
    system.append(ra.Spheroid(distance=1.))
    with pytest.raises(ValueError):
        g.propagate()
    with pytest.raises(ra.EngineError):
        ra.Engine(device=4096)
    big = ra.Spheroid(aspherics=[0.]*11, distance=1.)
    with pytest.raises(ValueError):
        pack_system(ra.System([ra.Spheroid(), big]), 5e-7, 1.)


def test_rccl_gather_single_rank_pipeline():
    """rt_comm_* / rt_gather_final with a one-rank communicator: exercises the
    RCCL load, communicator creation, the staged snapshot pipeline (two
    steps, so both parities and the event chain are used) and the root's
    [component][global ray] layout."""
    from rayopt_amd._lib import RT_Y, RT_T
    from rayopt_amd.distributed import split_gathered
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    y, u = _c3_rays(100003)
    g = gpu_trace(system, y, u, None, True)
    eng = g.engine
    with pytest.raises(ra.EngineError, match="rt_comm_init first"):
        eng.comm_info()
    eng.comm_init(eng.comm_unique_id(), 1, 0)
    # what the communicator says of itself (RCCL, not the stand-in)
    info = eng.comm_info()
    assert info["nranks_seen"] == 1 and info["rank_seen"] == 0
    assert info["rccl_version_code"] > 20000 and info["devices_visible"] >= 1
    assert all(l["device"] != eng.device for l in info["links"])
    counts = np.array([g.nrays], dtype=np.int64)
    d_dst = eng.scratch(g.nrays*3*8)
    L = len(system)
    for step in range(3):
        eng.trace(1, 0, True)
        eng.gather_final(RT_Y, L - 1, counts, 0, d_dst)
    eng.comm_sync()
    got = eng.copy_to_host(d_dst, g.nrays*3*8)
    parts = split_gathered(got, counts)
    assert np.array_equal(parts[0], np.asarray(g.y[-1]), equal_nan=True)
    eng.gather_final(RT_T, L - 2, counts, 0, d_dst)
    eng.comm_sync()
    got = eng.copy_to_host(d_dst, g.nrays*8)
    assert np.array_equal(got, np.asarray(g.t[-2]), equal_nan=True)
    with pytest.raises(ra.EngineError):
        eng.gather_final(RT_Y, L - 1, np.array([5], dtype=np.int64), 0, d_dst)


def test_i_rows_alias_u_only_where_identical():
    """i[j] is served from u[j-1] when neither element is tilted; the
    device pointers show which rows were materialised."""
    from rayopt_amd._lib import RT_U, RT_I
    system = ra.system_from_yaml(P.TORTURE)   # rotated: 1,2,3,5,6,7,8
    y, u = disc_bundle(4096, 9., 2., 5)
    g = gpu_trace(system, y, u, None, True)
    eng = g.engine
    rotated = [bool(e.rotated) for e in system]
    for j in range(1, len(system)):
        aliased = eng.device_ptr(RT_I, j) == eng.device_ptr(RT_U, j - 1)
        assert aliased == (not rotated[j] and not rotated[j - 1]), j
    assert eng.device_ptr(RT_I, 0) == eng.device_ptr(RT_U, 0)
    full = gpu_trace(system, y, u, None, True, alias_i=0)
    assert full.engine.device_ptr(RT_I, 4) != full.engine.device_ptr(RT_U, 3)
    assert np.array_equal(np.asarray(g.i), np.asarray(full.i), equal_nan=True)


def test_keep_rows_extension():
    """propagate(keep=...) stores only the chosen rows; they are bit-identical
    to the full trace, the others raise on access."""
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    y, u = _c3_rays(30011)
    full = gpu_trace(system, y, u, None, True)
    g = ra.GeometricTrace(system)
    g.rays_given(y, u)
    g.propagate(clip=True, keep=[-1])
    for a, b in ((g.y, full.y), (g.u, full.u), (g.i, full.i), (g.t, full.t)):
        assert np.array_equal(a[-1], b[-1], equal_nan=True)
        assert np.array_equal(a[0], b[0], equal_nan=True)
    assert np.array_equal(g.n, full.n)
    with pytest.raises(ra.EngineError, match="holds no data"):
        g.y[3]
    with pytest.raises(ra.EngineError, match="holds no data"):
        g.rms(i=5)
    assert np.isnan(g.rms()) == np.isnan(full.rms())
    with pytest.raises(ra.EngineError, match="seed row"):
        g.propagate(start=6, clip=True)
    # a sparse selection; i[6] must be materialised because row 5 is absent,
    # i[3] may alias u[2]
    g.propagate(clip=True, keep=[2, 3, 6, -1])
    for j in (2, 3, 6, 12):
        for a, b in ((g.y, full.y), (g.u, full.u), (g.i, full.i),
                     (g.t, full.t)):
            assert np.array_equal(a[j], b[j], equal_nan=True), j
    # restart from a kept row reproduces the tail
    g.propagate(start=7, clip=True)
    for a, b in ((g.y, full.y), (g.u, full.u), (g.i, full.i), (g.t, full.t)):
        assert np.array_equal(a[7:], b[7:], equal_nan=True)
    # and the default restores the reference behaviour
    g.propagate(clip=True)
    assert np.array_equal(np.asarray(g.y), np.asarray(full.y), equal_nan=True)
    assert np.array_equal(np.asarray(g.i), np.asarray(full.i), equal_nan=True)


def test_optimiser_as_caller():
    """examples/optimize_spot.py: the merit function (image row only + device
    rms) drives scipy's Nelder-Mead; the spot must shrink."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), "examples", "optimize_spot.py")
    spec = importlib.util.spec_from_file_location("optimize_spot", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    f0, f1, nfev = mod.main(nrays=200_000, verbose=False)
    assert f1 < 0.5*f0 and nfev > 20


def test_big_job_in_batches_example():
    """examples/big_job_in_batches.py: a job traced as several batches; the
    per-field statistics combined from the batches are those of the same
    rays traced as one batch."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), "examples", "big_job_in_batches.py")
    spec = importlib.util.spec_from_file_location("big_job_in_batches", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cnt, centroid, rms, traces = mod.main(total=600_000, batch=200_000,
                                          verbose=False)
    assert len(traces) == 3
    per = traces[0].nrays//len(mod.FIELDS)
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    g = ra.GeometricTrace(system)
    g.rays_fields(mod.FIELDS,
                  np.concatenate([mod.pupil_points(per, b) for b in range(3)]),
                  np.full(len(mod.FIELDS), P.DOUBLE_GAUSS_PUPIL_Z), 17.)
    g.propagate(clip=True)
    s = g.spot_stats(group_rays=3*per)
    assert 0 < (s[:, 0] < 3*per).sum()            # some bundles are clipped
    assert np.array_equal(cnt, s[:, 0])
    np.testing.assert_allclose(centroid, s[:, 1:3], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(rms, np.sqrt(s[:, 3]), rtol=1e-10)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 127, 129])
def test_ragged_small_batches(n):
    """N not a multiple of the 64-ray padding, down to a single ray."""
    system = ra.system_from_yaml(P.TORTURE)
    y, u = disc_bundle(n, 9., 2., n)
    g = gpu_trace(system, y, u, None, True)
    want, _ = oracle_trace(system, y, u, g.l, True)
    compare(g, want, 1, 9, RTOL_SPHERICAL, "n=%d" % n)
    assert g.y.shape == (9, n, 3)


def test_maximum_elements_and_aspheric_terms():
    """RT_MAX_SURFACES = 256 elements, RT_MAX_ASPH = 10 terms."""
    rng = np.random.default_rng(3)
    els = [{"material": 1.0}]
    for j in range(254):
        el = {"distance": 0.4, "radius": 6.0,
              "material": float(1.0 + 0.5*(j % 2))}
        if j % 3 == 0:
            el["roc"] = float(rng.choice([-1, 1])*rng.uniform(40, 300))
        if j % 50 == 7:
            el["aspherics"] = [float(rng.normal()*1e-4/6.**(2*i + 1))
                               for i in range(10)]
        els.append(el)
    els.append({"distance": 5.0, "radius": 50.})
    system = ra.System(elements=els, wavelengths=[587.56e-9])
    assert len(system) == 256
    y, u = disc_bundle(3000, 4., 1., 2)
    g = gpu_trace(system, y, u, None, True)
    want, ns = oracle_trace(system, y, u, g.l, True)
    compare(g, want, 1, 256, RTOL_ASPHERE, "L=256")
    assert np.array_equal(g.n[1:], ns[1:])
    with pytest.raises(ValueError):
        pack_system(ra.System(elements=els + [{"distance": 1.}]), 5e-7, 1.)


def test_huge_batch_64bit_indexing():
    """6*10^7 rays x 13 elements: 62 GB of results, element offsets beyond
    2^31 -- rays are built on the device, single rays are read back across
    all surfaces (rt_download_ray) and compared with the oracle."""
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    per_field = 10_000_000
    fields = np.c_[np.zeros(6), np.linspace(0, 1, 6)]
    rng = np.random.default_rng(1)
    r, phi = np.sqrt(rng.random(per_field)), 2*np.pi*rng.random(per_field)
    yp = np.c_[r*np.cos(phi), r*np.sin(phi)]
    g = ra.GeometricTrace(system)
    g.rays_fields(fields, yp, P.DOUBLE_GAUSS_PUPIL_Z, 17.)
    n = g.nrays
    assert n == 60_000_000 and 13*3*g.engine.ld > 2**31
    g.propagate(clip=True)
    from rayopt_amd._lib import RT_Y, RT_U, RT_I, RT_T
    picks = [0, 1, 63, 12_345_678, 35_791_394, 35_791_395, 47_721_858,
             n - 65, n - 1]
    cols = {w: np.array([g.engine.download_ray(w, k) for k in picks])
            for w in (RT_Y, RT_U, RT_I, RT_T)}
    y0, u0 = cols[RT_Y][:, 0], cols[RT_U][:, 0]
    want, _ = oracle_trace(system, y0, u0, g.l, True)
    for w, ref in zip((RT_Y, RT_U, RT_I, RT_T), want):
        got = np.moveaxis(cols[w], 0, 1)[1:]       # (L-1, rays[, 3])
        assert_parity(got, ref, RTOL_SPHERICAL, "huge")
    assert np.isfinite(cols[RT_T][:, 1:]).any()
    assert g.rms(i=1) > 0


@pytest.mark.parametrize("n,block", [(100_003, 0), (100_003, 4096),
                                     (5000, 256)])
def test_download_rays_is_a_strided_view_of_the_rows(n, block):
    """rt_download_rays (a sample of every row, gathered on the device)
    against the same rays cut out of the downloaded rows: plain layout and
    in blocks, served rows (i = u[j-1]), rows that were not stored."""
    from rayopt_amd._lib import RT_Y, RT_U, RT_I, RT_T
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    L = len(system)
    y, u = disc_bundle(n, 17., 9., 4, P.DOUBLE_GAUSS_PUPIL_Z)
    g = ra.GeometricTrace(system)
    if block:
        g.engine.set_option("block_rays", block)
    g.rays_given(y, u)
    g.propagate(clip=True)
    for ray0, stride, count in ((0, 1, 64), (3, 97, (n - 4)//97),
                                (n - 1, 1, 1), (0, n - 1, 2)):
        for which, rows in ((RT_Y, g.y), (RT_U, g.u), (RT_I, g.i),
                            (RT_T, g.t)):
            got = g.engine.download_rays(which, ray0, stride, count)
            want = np.asarray(rows[:])[:, ray0:ray0 + count*stride:stride]
            assert got.shape == want.shape
            assert np.array_equal(got, want, equal_nan=True), (which, ray0)
    g.propagate(clip=True, keep=[-1])
    got = g.engine.download_rays(RT_Y, 0, 7, 100)
    assert np.isnan(got[1:L - 1]).all() and np.isfinite(got[L - 1]).any()
    for bad in ((0, 1, n + 1), (-1, 1, 1), (5, n, 2), (0, 0, 1)):
        with pytest.raises(ra.EngineError):
            g.engine.download_rays(RT_Y, *bad)


def test_parity_sample_at_C5_size():
    """BASELINE configs[4] on ONE GPU: 10^8 rays built on the device (104 GB
    of results in 15 blocks).  Every 10^4-th ray -- 10^4 rays, spread over
    all blocks and all five field bundles -- is gathered on the device
    across all surfaces and compared with the oracle, which starts from the
    launch rays the device itself built (row 0 of the same sample): bit for
    bit, like every spherical system."""
    import digest_cases as dc
    from rayopt_amd._lib import RT_Y, RT_U, RT_I, RT_T
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    L = len(system)
    fields = np.c_[np.zeros(5), (0, .35, .5, .7, 1.)]
    per = 20_000_000
    g = ra.GeometricTrace(system)
    pts = dc.disc_points(per, 91)
    g.rays_fields(fields, pts, P.DOUBLE_GAUSS_PUPIL_Z, 17.)
    n = g.nrays
    assert n == 100_000_000
    g.propagate(clip=True)
    nb, bs, _ = g.engine.blocks()
    assert nb >= 14
    stride, count = 10_000, n//10_000
    cols = {w: g.engine.download_rays(w, 137, stride, count)
            for w in (RT_Y, RT_U, RT_I, RT_T)}
    rays = 137 + stride*np.arange(count)
    assert len(np.unique(rays//bs)) == nb          # every block is sampled
    assert len(np.unique(rays//per)) == 5          # and every bundle
    y0, u0 = cols[RT_Y][0], cols[RT_U][0]
    assert np.array_equal(cols[RT_I][0], u0) and not cols[RT_T][0].any()
    # the launch rays the device built at this size against the aiming oracle
    # (oracle/aim_numpy.py, pinned to the reference's System.aim by
    # tests/test_generate.py): the sampled rays of every bundle from their own
    # pupil points (generation at the largest shape was only
    # compared at 5000 points)
    from test_generate import oracle_rays
    a2 = 17.*np.array(((-1., -1.), (1., 1.)))
    for f in range(5):
        sel = rays//per == f
        with np.errstate(all="ignore"):
            yr, ur = oracle_rays(system, fields[f], pts[rays[sel] % per],
                                 P.DOUBLE_GAUSS_PUPIL_Z, a2)
        assert_parity(y0[sel][None], yr[None], 1e-13, "C5 launch y, field %d"
                      % f)
        assert_parity(u0[sel][None], ur[None], 1e-13, "C5 launch u, field %d"
                      % f)
    want, _ = oracle_trace(system, y0, u0, g.l, True)
    dead = 0
    for w, ref in zip((RT_Y, RT_U, RT_I, RT_T), want):
        got = cols[w][1:]
        assert np.array_equal(got, ref, equal_nan=True), w
        dead = max(dead, int(np.isnan(got[-1]).sum()))
    assert 0 < dead < count//5                     # some rays vignette


def test_element_level_methods():
    """Spheroid.propagate / .intercept on the device vs the oracle's
    element_propagate (rayopt/elements.py:306-315, 477-501)."""
    rng = np.random.default_rng(8)
    y = np.c_[rng.uniform(-4, 4, (500, 2)), rng.uniform(-3, -1, 500)]
    u = rng.normal(size=(500, 3))*[.05, .05, 0] + [0, 0, 1.]
    u /= np.sqrt(np.square(u).sum(1))[:, None]
    for kw in (dict(roc=30., material=1.6, radius=3.5),
               dict(roc=-40., conic=-1.7, material="mirror", radius=5.),
               dict(roc=25., conic=.2, aspherics=[0, 1e-5, -2e-7],
                    material=1.5, radius=3.),
               dict(material=1.33, radius=2., angles=(.2, 0, 0))):
        el = ra.Spheroid(**kw)
        table, ns = pack_system([ra.Spheroid(), el], 5.5e-7, 1.1)
        table["offset"][1] = 0.
        table["flags"][1] &= ~np.uint32(1)
        for clip in (True, False):
            yo, uo, to = tn.element_propagate(table[1], y, u, clip)
            yg, ug, ng, tg = el.propagate(y, u, 1.1, 5.5e-7, clip)
            rtol = RTOL_ASPHERE if "aspherics" in kw else RTOL_SPHERICAL
            assert_parity(yg[None], yo[None], rtol, "el.y")
            assert_parity(ug[None], uo[None], rtol, "el.u")
            assert_parity(tg[None], to[None], rtol, "el.t")
            assert ng == ns[1]
        s = el.intercept(y, u)
        assert_parity((s*1.1)[None], to[None], 1e-12, "el.intercept")


def test_config_c2_three_wavelengths_one_launch():
    """BASELINE config C2 literally: 10^6 rays x 3 wavelengths in ONE trace
    (ray groups, one surface table per wavelength); every group is
    bit-identical to the same rays traced alone at that wavelength."""
    ls = [587.56e-9, 656.27e-9, 486.13e-9]
    p = 10**6 - 10**6 % 64
    y, u = disc_bundle(p, 5.5, 5., 0)
    # a system whose indices depend on the wavelength: Abbe models fitted to
    # the catalogue values of the Cooke fixture
    idx = P.COOKE_INDICES
    def abbe(key):
        nd, nf, nc = (idx[l][key] for l in (587.56e-9, 486.13e-9, 656.27e-9))
        return "%r/%r" % (nd, (nd - 1)/(nf - nc))
    text = P.COOKE % dict(air=1.0, sk16=0., f2=0.)
    text = text.replace("material: 0.0,", "material: @,")
    parts = text.split("@")
    glasses = [abbe("sk16"), abbe("f2"), abbe("sk16")]
    text = "".join(a + b for a, b in zip(parts, glasses + [""]))
    system = ra.system_from_yaml(text)
    g = ra.GeometricTrace(system)
    g.rays_given(y, u, ls)
    g.propagate(clip=True)
    assert g.nrays == 3*p and g.n.shape == (3, 9)
    ms = g.kernel_ms()
    for k, l in enumerate(ls):
        single = gpu_trace(system, y, u, l, True)
        sl = slice(k*p, (k + 1)*p)
        for j in (1, 4, 8):
            for a, b in ((g.y, single.y), (g.u, single.u), (g.i, single.i)):
                assert np.array_equal(np.asarray(a[j])[sl], np.asarray(b[j]),
                                      equal_nan=True)
            assert np.array_equal(np.asarray(g.t[j])[sl],
                                  np.asarray(single.t[j]), equal_nan=True)
        assert np.array_equal(g.n[k], single.n)
    assert g.n[0, 1] != g.n[1, 1] and ms > 0
    want, _ = oracle_trace(system, y[:6400], u[:6400], ls[2], True)
    sub = ra.GeometricTrace(system)
    sub.rays_given(y[:6400], u[:6400], ls)
    sub.propagate(clip=True)
    for rows, b in zip((sub.y, sub.u, sub.i, sub.t), want):
        assert_parity(np.asarray(rows[1:])[:, 2*6400:], b, RTOL_SPHERICAL,
                      "group 2")


@pytest.mark.parametrize("n", [100_000, 200_000, 1_400_000, 1_500_000,
                               3_000_001])
def test_upload_download_round_trip(n):
    """Row 0 read back equals what rays_given was handed, across the direct
    and the pinned double-buffered staging paths (one chunk, chunk boundary,
    several chunks) -- y and u share the staging buffers back to back."""
    rng = np.random.default_rng(n)
    y = rng.normal(size=(n, 3))
    u = rng.normal(size=(n, 3))
    g = ra.GeometricTrace(ra.system_from_yaml(P.SINGLET))
    for rep in range(2):
        g.rays_given(y, u)
        assert np.array_equal(g.y[0], y) and np.array_equal(g.u[0], u)
        assert np.array_equal(g.i[0], u)
        y, u = u, y


def test_many_contexts_lifecycle():
    """Analysis-style use: dozens of small traces alive at once (one context
    each), and hundreds created and dropped in sequence."""
    import gc
    system = ra.system_from_yaml(P.COOKE % P.COOKE_INDICES[587.56e-9])
    y, u = disc_bundle(1000, 5., 3., 1)
    alive = []
    for k in range(40):
        g = gpu_trace(system, y, u, None, bool(k % 2))
        alive.append(g)
    ref = np.asarray(alive[0].y[-1])
    for g in alive[2::2]:
        assert np.array_equal(np.asarray(g.y[-1]), ref, equal_nan=True)
    del alive
    gc.collect()
    for k in range(300):
        g = gpu_trace(system, y, u, None, False)
        assert np.isfinite(g.rms()) or True
        del g
    gc.collect()
    big = gpu_trace(ra.system_from_yaml(P.DOUBLE_GAUSS), *_c3_rays(2_000_000),
                    None, True)
    assert big.y.shape == (13, 2_000_000, 3)


def test_negative_index_medium():
    from test_oracle_golden import NEGATIVE_INDEX
    system = ra.system_from_yaml(NEGATIVE_INDEX)
    y, u = disc_bundle(20000, 10., 3., 1)
    g = gpu_trace(system, y, u, None, True)
    want, _ = oracle_trace(system, y, u, g.l, True)
    compare(g, want, 1, 4, RTOL_SPHERICAL, "negative index")
    assert np.isfinite(np.asarray(g.y[-1])).any()


def test_nan_and_inf_inputs_propagate_like_numpy():
    """NaN / inf / zero-direction launch data: in-band failure exactly where
    numpy produces it (the reference has no input validation)."""
    system = ra.system_from_yaml(P.TORTURE)
    y, u = disc_bundle(640, 8., 2., 9)
    y[::7, 0] = np.nan
    u[3::11, 1] = np.nan
    y[5::13, 2] = np.inf
    u[2::17] = 0.
    u[4::19, 2] = -u[4::19, 2]
    y[6::23] = 1e300
    for clip in (True, False):
        g = gpu_trace(system, y, u, None, clip)
        with np.errstate(all="ignore"):
            want, _ = oracle_trace(system, y, u, g.l, clip)
        compare(g, want, 1, 9, RTOL_SPHERICAL, "nan inputs")
    a = ra.system_from_yaml(P.ASPHERE_PHONE)
    y, u = disc_bundle(640, 0.6, 10., 9)
    y[::7, 1] = np.nan
    u[2::17] = 0.
    y[5::13, 2] = -np.inf
    g = gpu_trace(a, y, u, None, True)
    with np.errstate(all="ignore"):
        want, _ = oracle_trace(a, y, u, g.l, True)
    compare(g, want, 1, 9, RTOL_ASPHERE, "nan inputs asphere")


def test_full_size_every_ray_against_c_oracle():
    """BASELINE C3 at full size, EVERY ray and row: the device result equals
    the plain-C oracle (oracle/trace_c.c, itself bit-identical to the
    reference's goldens) bit for bit -- 1.2*10^8 ray-surface ops, 10 values
    each."""
    from oracle import build_c
    n = 10**7
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    y, u = _c3_rays(n)
    g = gpu_trace(system, y, u, None, True)
    table, ns = pack_system(system, g.l, g.n[0])
    Y, U, I, T = build_c.propagate(table, y, u, clip=True)
    for j in range(1, len(system)):
        for rows, ref in ((g.y, Y), (g.u, U), (g.i, I)):
            assert np.array_equal(rows[j], ref[j - 1], equal_nan=True), j
        assert np.array_equal(g.t[j], T[j - 1], equal_nan=True), j
    assert np.array_equal(g.n[1:], ns[1:])


def test_full_size_asphere_every_ray_against_c_oracle():
    from oracle import build_c
    n = 4*10**6
    system = ra.system_from_yaml(P.ASPHERE_PHONE)
    y, u = disc_bundle(n, 0.62, 20., 5)
    y[:, 1] -= 0.5*np.tan(np.radians(20.))
    g = gpu_trace(system, y, u, None, True)
    table, ns = pack_system(system, g.l, g.n[0])
    want = build_c.propagate(table, y, u, clip=True)
    # 4*10^6 rays through six Newton-solved aspheres: every value of every
    # row equal to the C oracle's (itself bit-identical to the reference's
    # asphere goldens), bit for bit
    for rows, ref in zip((g.y, g.u, g.i, g.t), want):
        assert np.array_equal(np.asarray(rows[1:]), ref, equal_nan=True)


# -- repeated traces of one seed, new seeds, partial re-propagation ------------

def _rows(g, which=("y", "u", "t")):
    return [np.array(np.asarray(getattr(g, k))) for k in which]


@pytest.mark.gpu
@pytest.mark.parametrize("n", [10_000, 1_000_037])
def test_repeated_traces_are_identical_and_follow_new_seeds(n):
    """Re-tracing the same seed is bit-identical every time (and equal to the
    oracle) under every kernel variant; new rays or a row uploaded through
    the C ABI take effect."""
    from oracle import build_c
    system = ra.system_from_yaml(ra.prescriptions.DOUBLE_GAUSS)
    y, u = ra.bundles.disc_bundle(n, 17., 10., 3,
                                  ra.prescriptions.DOUBLE_GAUSS_PUPIL_Z)
    g = ra.GeometricTrace(system)
    g.rays_given(y, u)
    table, _ = pack_system(system, g.l, g.n[0])
    want = build_c.propagate(table, y, u, clip=True)
    first = None
    for rep in range(4):
        g.propagate(clip=True)
        got = _rows(g, ("y", "u", "i", "t"))
        for a, b in zip(got, want):
            assert np.array_equal(a[1:], b, equal_nan=True), rep
        first = first or got
    for variant in (dict(alias_i=0), dict(compact=2),
                    dict(compact=2, compact_every=1)):
        for k, v in variant.items():
            g.engine.set_option(k, v)
        for rep in range(3):
            g.propagate(clip=True)
            assert np.array_equal(np.asarray(g.y[-1]), first[0][-1],
                                  equal_nan=True), (variant, rep)
        for k, v in dict(alias_i=1, compact=0, compact_every=4).items():
            g.engine.set_option(k, v)
    # new rays of the same count
    y2, u2 = ra.bundles.disc_bundle(n, 12., -6., 9,
                                    ra.prescriptions.DOUBLE_GAUSS_PUPIL_Z)
    g.rays_given(y2, u2)
    want2 = build_c.propagate(table, y2, u2, clip=True)
    for rep in range(3):
        g.propagate(clip=True)
        assert np.array_equal(np.asarray(g.y)[1:], want2[0], equal_nan=True)
    # a row uploaded behind the engine's back (C ABI) is seen too
    g.engine.upload_row(0, 0, np.ascontiguousarray(y.T))     # RT_Y, row 0
    g.engine.upload_row(1, 0, np.ascontiguousarray(u.T))     # RT_U, row 0
    g.engine.trace(1, 0, True)
    assert np.array_equal(np.asarray(g.engine.download(0, 12, 13))[0].T,
                          want[0][-1], equal_nan=True)


@pytest.mark.gpu
def test_repeated_partial_propagation():
    """propagate(start=k) seeds from row k-1, repeatedly; a full trace that
    rewrites that row is picked up."""
    system = ra.system_from_yaml(ra.prescriptions.DOUBLE_GAUSS)
    y, u = ra.bundles.disc_bundle(300_000, 17., 10., 5,
                                  ra.prescriptions.DOUBLE_GAUSS_PUPIL_Z)
    g = ra.GeometricTrace(system)
    g.rays_given(y, u)
    g.propagate(clip=True)
    full = _rows(g)
    for rep in range(3):                 # seed row 5, three times
        g.propagate(start=6, clip=True)
        for a, b in zip(_rows(g), full):
            assert np.array_equal(a, b, equal_nan=True)
    system[3].curvature *= 1.01          # rows 3.. change, row 5 with them
    g.propagate(clip=True)
    changed = _rows(g)
    assert not np.array_equal(changed[0][5], full[0][5], equal_nan=True)
    for rep in range(3):
        g.propagate(start=6, clip=True)
        for a, b in zip(_rows(g), changed):
            assert np.array_equal(a, b, equal_nan=True)


@pytest.mark.gpu
def test_repeated_traces_with_wavelength_groups():
    system = ra.system_from_yaml(ra.prescriptions.COOKE % dict(
        air=1.0, sk16="1.62041/60.32", f2="1.62004/36.37"))
    y, u = ra.bundles.disc_bundle(64*500, 5., 4., 2)
    ls = system.wavelengths
    g = ra.GeometricTrace(system)
    g.rays_given(y, u, l=ls)
    ref = None
    for rep in range(4):
        g.propagate(clip=True)
        rows = _rows(g)
        if ref is None:
            ref = rows
        for a, b in zip(rows, ref):
            assert np.array_equal(a, b, equal_nan=True)
    for w, l in enumerate(ls):
        h = ra.GeometricTrace(system)
        h.rays_given(y, u, l=l)
        h.propagate(clip=True)
        assert np.array_equal(np.asarray(h.y[-1]),
                              ref[0][-1][w*len(y):(w + 1)*len(y)],
                              equal_nan=True)


# -- u[j] served from i[j] for unclipped traces (RT_F_SKIP_U) ----------------

@pytest.mark.gpu
def test_u_rows_of_unbending_elements_are_served_not_stored():
    """Unclipped trace: u at the stop and the image is i bit for bit and is
    not written; every consumer still sees the reference's values -- full
    download, toggling clip, keep=[-1], partial re-propagation on either
    side of such a row, rows beyond `stop` keeping their old content."""
    from oracle import build_c
    system = ra.system_from_yaml(ra.prescriptions.DOUBLE_GAUSS)
    L = len(system)
    stop = system.stop
    y, u = ra.bundles.disc_bundle(200_003, 17., 10., 3,
                                  ra.prescriptions.DOUBLE_GAUSS_PUPIL_Z)
    g = ra.GeometricTrace(system)
    g.rays_given(y, u)
    table, _ = pack_system(system, g.l, g.n[0])

    def check(clip, rows=slice(1, None)):
        want = build_c.propagate(table, y, u, clip=clip)
        for name, b in zip("yuit", want):
            a = np.asarray(getattr(g, name))[1:]
            assert np.array_equal(a[rows.start - 1:], b[rows.start - 1:],
                                  equal_nan=True), (name, clip)
        return want

    for clip in (False, True, False, False, True):
        g.propagate(clip=clip)
        check(clip)
    # device pointers of the served rows resolve to stored data
    g.propagate(clip=False)
    eng = g.engine
    assert eng.device_ptr(1, stop) == eng.device_ptr(2, stop)   # U -> I
    assert eng.device_ptr(1, stop) == eng.device_ptr(1, stop - 1)
    for alias in (0, 1):
        eng.set_option("alias_i", alias)
        g.propagate(clip=False)
        old = check(False)
    # image row only
    g.propagate(clip=False, keep=[-1])
    assert np.array_equal(np.asarray(g.u[-1]), old[1][-1], equal_nan=True)
    assert np.array_equal(np.asarray(g.i[-1]), old[2][-1], equal_nan=True)
    # seed from a served row: re-trace the rear half from the stop
    g.propagate(clip=False)
    g.propagate(start=stop + 1, clip=False)
    check(False)
    # rewrite the front half of a CHANGED system up to (not including) the
    # stop: rows from the stop on keep the old trace, as in the reference
    system[2].curvature *= 1.02
    g.propagate(stop=stop, clip=False)
    table2, _ = pack_system(system, g.l, g.n[0])
    new = build_c.propagate(table2, y, u, clip=False)
    for k, name in enumerate("yui"):
        a = np.asarray(getattr(g, name))[1:]
        assert np.array_equal(a[:stop - 1], new[k][:stop - 1], equal_nan=True)
        assert np.array_equal(a[stop - 1:], old[k][stop - 1:],
                              equal_nan=True), name
        assert not np.array_equal(new[k][stop - 2], old[k][stop - 2],
                                  equal_nan=True)
    assert L - 1 > stop


@pytest.mark.gpu
def test_opd_reference_row_served_from_i():
    """opd(after=stop): the kernel reads U[after] at its natural address, so
    a served row is given its own copy first; same numbers as with every row
    materialised."""
    system = ra.system_from_yaml(ra.prescriptions.DOUBLE_GAUSS)
    y, u = ra.bundles.disc_bundle(50_000, 10., 0., 3,
                                  ra.prescriptions.DOUBLE_GAUSS_PUPIL_Z)
    out = []
    for alias in (1, 0):
        g = ra.GeometricTrace(system)
        g.engine.set_option("alias_i", alias)
        g.rays_given(y, u)
        g.propagate(clip=False)
        out.append(g.opd(radius=80., after=system.stop, resample=0))
        # the row is still right after having been given its own storage
        want = np.asarray(g.i[system.stop])
        assert np.array_equal(np.asarray(g.u[system.stop]), want,
                              equal_nan=True)
    for a, b in zip(*out):
        assert np.array_equal(a, b, equal_nan=True)


@pytest.mark.gpu
def test_uploaded_rows_do_not_leak_into_rows_served_from_them():
    """y, u, i are independent arrays in the reference: replacing a row of U
    (or I) through the C ABI must not change the row of I (or U) that was
    being served from it."""
    system = ra.system_from_yaml(ra.prescriptions.DOUBLE_GAUSS)
    stop = system.stop
    y, u = ra.bundles.disc_bundle(4096, 15., 5., 2,
                                  ra.prescriptions.DOUBLE_GAUSS_PUPIL_Z)
    g = ra.GeometricTrace(system)
    g.rays_given(y, u)
    g.propagate(clip=False)
    eng = g.engine
    U, I = 1, 2
    old_u = np.array(eng.download(U, 0, len(system)))
    old_i = np.array(eng.download(I, 0, len(system)))
    assert np.array_equal(old_i[4], old_u[3])            # served from U[3]
    assert np.array_equal(old_u[stop], old_i[stop])      # served from I[stop]
    junk = np.full((3, 4096), 7.25)
    eng.upload_row(U, 3, junk)
    assert np.array_equal(eng.download(U, 3, 4)[0], junk)
    assert np.array_equal(eng.download(I, 4, 5)[0], old_i[4])
    eng.upload_row(I, stop, junk)
    assert np.array_equal(eng.download(I, stop, stop + 1)[0], junk)
    assert np.array_equal(eng.download(U, stop, stop + 1)[0], old_u[stop])
    # and the next row, served from U[stop], still shows the old direction
    assert np.array_equal(eng.download(I, stop + 1, stop + 2)[0],
                          old_i[stop + 1])


@pytest.mark.gpu
def test_system_variants_in_one_trace_gpu():
    from test_host_model import _variants_checks
    _variants_checks(lambda s: ra.GeometricTrace(s))


@pytest.mark.gpu
def test_two_thousand_variants_one_launch():
    """A tolerancing run: 2000 perturbed triplets x 192 rays in one launch
    (one surface table per variant); spot statistics per variant against
    the oracle on a sample of them."""
    import copy
    from oracle import build_c
    from oracle import consumers_numpy as cn
    base = ra.system_from_yaml(ra.prescriptions.cooke())
    rng = np.random.default_rng(3)
    variants = []
    for v in range(2000):
        s = copy.deepcopy(base)
        for el in s[1:-1]:
            el.curvature *= 1 + 2e-3*rng.standard_normal()
        variants.append(s)
    y, u = ra.bundles.disc_bundle(192, 4., 8., 1)
    g = ra.GeometricTrace(base)
    g.rays_variants(y, u, variants)
    g.propagate(clip=True, keep=[-1])
    stats = g.spot_stats()
    assert stats.shape == (2000, 6) and g.kernel_ms() < 5.
    rows = np.asarray(g.y[-1]).reshape(2000, 192, 3)
    for v in (0, 1, 777, 1999):
        table, _ = pack_system(variants[v], g.l, g.n[v, 0])
        want = build_c.propagate(table, y, u, clip=True)[0][-1]
        assert np.array_equal(rows[v], want, equal_nan=True)
        ref = cn.spot_stats(want, 192)[0]
        assert stats[v, 0] == ref[0]
        np.testing.assert_allclose(stats[v, 1:5], ref[1:5], rtol=1e-9)
    assert np.std(np.sqrt(stats[:, 3])) > 0


@pytest.mark.gpu
def test_grouped_partial_propagation_gpu():
    from test_host_model import _grouped_partial_checks
    _grouped_partial_checks(lambda s: ra.GeometricTrace(s))


# -- clipped-ray compaction (rt_set_option "compact") --------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("scale", [1., 1.6])
@pytest.mark.parametrize("keep", [None, [-1], [3, 7, -1]])
def test_compacting_kernel_gives_the_plain_kernels_results(scale, keep):
    """Dead rays retired, survivors packed into fewer wavefronts: every kept
    row identical to the plain kernel's, bit for bit, NaN masks included;
    over-filled bundle (most rays vignette at different elements) and the
    nominal one; all rows, the image row, a few rows."""
    from rayopt_amd import prescriptions as P
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    fields = [f*P.DOUBLE_GAUSS_FIELD_DEG for f in (0, .35, .5, .7, 1.)]
    y, u = ra.bundles.multi_field_bundle(200_003, 17.*scale, fields, 5,
                                         P.DOUBLE_GAUSS_PUPIL_Z)
    y[::1000] = np.nan                      # dead on arrival
    L = len(system)
    rows = range(1, L) if keep is None else [range(L)[k] for k in keep]
    g = ra.GeometricTrace(system)
    g.rays_given(y, u)
    g.propagate(clip=True, keep=keep)
    want = {(name, j): np.array(getattr(g, name)[j])
            for name in "yuit" for j in rows}
    g.engine.set_option("compact", 2 if keep is None else 1)
    try:
        g.propagate(clip=True, keep=keep)
        for (name, j), ref in want.items():
            got = np.asarray(getattr(g, name)[j])
            assert np.array_equal(got, ref, equal_nan=True), (name, j)
        dead = np.isnan(want["u", L - 1][:, 0]).mean()
        assert dead > (.5 if scale > 1 else .001)
    finally:
        g.engine.set_option("compact", 0)


@pytest.mark.gpu
def test_compacting_kernel_with_ray_groups_and_aspheres():
    """Groups with their own surface table (three wavelengths, tiles never
    straddle a group) and the Newton path under compaction."""
    from rayopt_amd import prescriptions as P
    system = ra.system_from_yaml(P.ASPHERE_PHONE)
    y, u = ra.bundles.disc_bundle(25_600, 0.9, 20., 3)   # overfills the stop
    y[:, 1] -= 0.5*np.tan(np.radians(20.))
    g = ra.GeometricTrace(system)
    ls = [587.56e-9, 486.13e-9, 656.27e-9]
    g.rays_given(y, u, l=ls)
    g.propagate(clip=True, keep=[-1])
    want = np.array(g.y[-1])
    assert .2 < np.isnan(want[:, 0]).mean() < .95
    g.engine.set_option("compact", 1)
    try:
        g.propagate(clip=True, keep=[-1])
        assert np.array_equal(np.asarray(g.y[-1]), want, equal_nan=True)
    finally:
        g.engine.set_option("compact", 0)


def test_replacing_a_launch_row_leaves_the_rows_it_served_alone():
    """i[0] = u[0] at seeding time is a COPY in the reference
    (rayopt/geometric_trace.py:67); on the device I[0] is served from U[0]
    until U[0] is replaced (rt_upload_row) -- then it must keep the old
    directions.  Found by tests/tools/fuzz_state.py."""
    from rayopt_amd._lib import RT_U
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    y, u = _c3_rays(1000)
    g = ra.GeometricTrace(system)
    g.rays_given(y, u)
    other = np.ascontiguousarray(u[::-1].T)
    g.engine.upload_row(RT_U, 0, other)
    for rows in (g.y, g.u, g.i, g.t):
        rows.invalidate(0, g.length)
    assert np.array_equal(np.asarray(g.u[0]), other.T)
    assert np.array_equal(np.asarray(g.i[0]), u)
    g.propagate(clip=True)
    want, _ = oracle_trace(system, y, other.T, g.l, True)
    compare(g, want, 1, 13, RTOL_SPHERICAL, "replaced u[0]")
    assert np.array_equal(np.asarray(g.i[0]), u)
