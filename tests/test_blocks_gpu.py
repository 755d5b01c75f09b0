"""Batches in BLOCKS (csrc/rt_lay.h): a large batch is cut into blocks of
rays, each with its own planes, so that the rows one trace writes at once do
not lie further apart than the device's address translation likes (10^8 rays:
0.72 of the HBM spec as one block, 0.84 in blocks).  The blocks are an
addressing matter: with "block_rays" small enough to cut test-sized batches
up, every answer of the engine must be the one the plain layout gives -- rows
bit for bit, reductions to their usual tolerance."""
import numpy as np
import pytest

import rayopt_amd as ra
from rayopt_amd import prescriptions as P
from rayopt_amd._lib import RT_Y, RT_U, RT_I, RT_T
from rayopt_amd.bundles import disc_bundle

pytestmark = pytest.mark.gpu


def rows_of(g):
    return [np.array(np.asarray(r[:])) for r in (g.y, g.u, g.i, g.t)]


def pair(system, block, **options):
    """The same trace object twice: plain layout, and in blocks."""
    a = ra.GeometricTrace(system, **options)
    b = ra.GeometricTrace(system, **options)
    a.engine.set_option("block_rays", 2**31 - 1)   # (whatever the environment)
    b.engine.set_option("block_rays", block)
    return a, b


def same(x, y):
    return np.array_equal(x, y, equal_nan=True)


@pytest.mark.parametrize("n,block", [(100_003, 4096), (5000, 256),
                                     (70_000, 33_000), (257, 256),
                                     (1_000_000, 300_000)])
def test_rows_are_the_same_bits(n, block):
    """Host-seeded rays through the tilted torture system (i rows stored and
    served), clip on and off, partial ranges, one ray's column, a row
    uploaded in between."""
    system = ra.system_from_yaml(P.TORTURE)
    L = len(system)
    y, u = disc_bundle(n, 12., 2., 5)
    a, b = pair(system, block)
    for g in (a, b):
        g.rays_given(y, u)
        g.propagate(clip=True)
    nb, bs, bts = b.engine.blocks()
    assert a.engine.blocks() == (1, a.engine.ld, 0)
    assert nb == -(-n//bs) and nb > 1 and bs % 256 == 0 and bts == 10*L*bs
    assert b.engine.ld == nb*bs >= n
    for x, z in zip(rows_of(a), rows_of(b)):
        assert same(x, z)
    for ray in (0, bs - 1, bs, n - 1, n//2):
        for which in (RT_Y, RT_U, RT_I, RT_T):
            assert same(a.engine.download_ray(which, ray),
                        b.engine.download_ray(which, ray))
    # a partial re-trace from rows uploaded by the host, unclipped
    rng = np.random.default_rng(n)
    yy = np.asarray(a.y[3]).T.copy()
    yy[:2] += 1e-3*rng.standard_normal((2, n))
    for g in (a, b):
        g.engine.upload_row(RT_Y, 3, yy)
        g.engine.trace(4, 0, False)
        for r in (g.y, g.u, g.i, g.t):
            r.invalidate(3, L)
    for x, z in zip(rows_of(a), rows_of(b)):
        assert same(x, z)


@pytest.mark.parametrize("chunks", [2, 3, 7])
def test_pieces_of_a_step_cross_the_blocks(chunks):
    """rt_trace_chunk windows begin and end anywhere in any block."""
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    n = 50_001
    y, u = disc_bundle(n, 17., 9., 3, P.DOUBLE_GAUSS_PUPIL_Z)
    a, b = pair(system, 4096)
    a.rays_given(y, u)
    a.propagate(clip=True)
    b.rays_given(y, u)
    b.propagate(clip=True, chunks=chunks)
    assert b.engine.blocks()[0] > 1
    for x, z in zip(rows_of(a), rows_of(b)):
        assert same(x, z)


def test_generated_batches_groups_kept_rows_and_compaction():
    """Rays built on the device (first trace and re-trace), several
    wavelengths in one launch, kept-row subsets, the compacting kernel."""
    system = ra.system_from_yaml(P.COOKE % dict(
        air=1.0, sk16="1.62041/60.32", f2="1.62004/36.37"))
    ls = system.wavelengths
    fields = np.c_[np.zeros(3), (0., .6, 1.)]
    a, b = pair(system, 5000)
    for g in (a, b):
        g.rays_points(fields, wavelength=ls, nrays=2000,
                      distribution="hexapolar", clip=True)
    assert a.nrays == b.nrays > 15_000
    assert b.engine.blocks()[0] > 1
    for x, z in zip(rows_of(a), rows_of(b)):
        assert same(x, z)
    for g in (a, b):
        g.propagate(clip=True, keep=[2, -1])
    for j in (2, -1):
        assert same(np.asarray(a.y[j]), np.asarray(b.y[j]))
        assert same(np.asarray(a.t[j]), np.asarray(b.t[j]))
    np.testing.assert_allclose(b.rms_fields(), a.rms_fields(), rtol=1e-12)
    assert np.array_equal(a.spot_stats()[..., 0], b.spot_stats()[..., 0])
    # the compacting kernel on an over-filled bundle, image row only
    s2 = ra.system_from_yaml(P.DOUBLE_GAUSS)
    y, u = disc_bundle(40_000, 30., 14., 9, P.DOUBLE_GAUSS_PUPIL_Z)
    c, d = pair(s2, 4096, compact=2)
    for g in (c, d):
        g.rays_given(y, u)
        g.propagate(clip=True, keep=[-1])
    assert d.engine.blocks()[0] > 1
    assert 0.2 < np.isnan(np.asarray(c.y[-1])[:, 0]).mean() < 0.98
    assert same(np.asarray(c.y[-1]), np.asarray(d.y[-1]))
    assert same(np.asarray(c.u[-1]), np.asarray(d.u[-1]))


@pytest.mark.parametrize("points,turn,block", [
    (256*37, 256, 0), (256*37, 1024, 4096), (256*37, 3000, 0),
    (256*8, 256, 768), (256*37 + 64, 512, 0), (256*12, 256*12, 0)])
def test_bundles_over_a_large_pupil_go_in_turns(points, turn, block):
    """A generated batch of several bundles over the same pupil points is
    traced in TURNS when the points outgrow the Infinity Cache (rt_gen_wg,
    csrc/rt_lay.h: 128 MB of them; "turn_points" forces it at test sizes):
    the order in which workgroups take the rays, nothing else -- every row of
    the first trace (which also builds row 0) and of a re-trace (which builds
    the rays again in registers) is the one the ray order gives, with the
    batch in blocks as well, with several wavelengths in one launch, and
    where the bundles are no whole workgroups (then the order stays)."""
    from bench_legs import FIELD_FRACTIONS, BUNDLE_RADIUS
    import digest_cases as dc
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    nf = len(FIELD_FRACTIONS)
    pts = dc.disc_points(points, 5)
    fields = np.c_[np.zeros(nf), FIELD_FRACTIONS]
    for ls in (None, [587.56e-9, 486.13e-9]):
        got = []
        for t in (-1, turn):
            g = ra.GeometricTrace(system)
            g.engine.set_option("turn_points", t)
            g.engine.set_option("block_rays", block or 2**31 - 1)
            if ls is None:
                g.rays_fields(fields, pts, P.DOUBLE_GAUSS_PUPIL_Z,
                              BUNDLE_RADIUS)
            else:
                W = len(ls)
                g.rays_fields(fields, pts, [P.DOUBLE_GAUSS_PUPIL_Z]*W,
                              [BUNDLE_RADIUS]*W, l=ls)
            g.propagate(clip=True)
            first = rows_of(g)
            g.propagate(clip=True)         # the re-trace
            for r in (g.y, g.u, g.i, g.t):
                r.invalidate(1, len(system))
            again = rows_of(g)
            assert bool(block) == (g.engine.blocks()[0] > 1)
            got.append((first, again))
        for x, z in zip(got[0][0], got[1][0]):
            assert same(x, z)
        for x, z in zip(got[0][1], got[1][1]):
            assert same(x, z)
        for x, z in zip(got[0][0], got[0][1]):
            assert same(x, z)
        assert np.isfinite(got[0][0][0][-1]).mean() > .5


@pytest.mark.parametrize("n,block", [(100_003, 4096), (9_999, 512)])
def test_reductions_in_blocks(n, block):
    """rms / refocus / rmax / spot statistics / opd over a batch in blocks
    against the plain layout and numpy: one-pass and two-pass kernels,
    weights, bundles that straddle blocks."""
    from oracle import consumers_numpy as cn
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    L = len(system)
    y, u = disc_bundle(n, 12., 9., 21, P.DOUBLE_GAUSS_PUPIL_Z)
    rng = np.random.default_rng(n)
    w = rng.random(n) + .1
    a, b = pair(system, block)
    for weights in (None, w):
        for g in (a, b):
            g.rays_given(y, u, None, weights, n//2)
            g.propagate(clip=False)
        assert b.engine.blocks()[0] > 1
        Y, I = np.asarray(a.y[-1]), np.asarray(a.i[-1])
        assert same(Y, np.asarray(b.y[-1]))
        for one in (1, 0):
            b.engine.set_option("consumers_one_pass", one)
            assert b.rms() == pytest.approx(cn.rms(Y, weights), rel=1e-12)
            assert b.rms(ref=n//2) == pytest.approx(
                cn.rms(Y, weights, n//2), rel=1e-12)
            assert b.rms(ref=n - 1) == pytest.approx(
                cn.rms(Y, weights, n - 1), rel=1e-12)
            assert b.rms(i=3) == pytest.approx(a.rms(i=3), rel=1e-12)
            assert b.engine.refocus_shift(L - 1) == pytest.approx(
                cn.refocus_shift(Y, I, weights), rel=1e-9)
        b.engine.set_option("consumers_one_pass", 1)
        assert b.engine.row_rmax(L - 1) == a.engine.row_rmax(L - 1)
        for per in (n, 7) if n % 7 == 0 else (n,):
            sa = a.engine.spot_stats(L - 1, per, n//per)
            sb = b.engine.spot_stats(L - 1, per, n//per)
            assert np.array_equal(sa[:, 0], sb[:, 0])
            np.testing.assert_allclose(sb, sa, rtol=1e-9)
        xa, ya, ta = a.opd(radius=100., resample=0)
        xb, yb, tb = b.opd(radius=100., resample=0)
        assert same(xa, xb) and same(ya, yb) and same(ta, tb)
    # groups of 1429 rays (7 x 1429 = 10003; x 10 = 100030): bundles that
    # begin and end inside blocks
    m = 1429*(n//1429)
    y, u = disc_bundle(m, 17.5, 12., 7, P.DOUBLE_GAUSS_PUPIL_Z)  # vignetted
    for g in (a, b):
        g.rays_given(y, u)
        g.propagate(clip=True)
    sa = a.engine.spot_stats(L - 1, 1429, m//1429)
    sb = b.engine.spot_stats(L - 1, 1429, m//1429)
    assert np.array_equal(sa[:, 0], sb[:, 0]) and (sa[:, 0] < 1429).any()
    np.testing.assert_allclose(sb[:, 1:], sa[:, 1:], rtol=1e-9)


def test_the_automatic_choice():
    """One block up to 8.5 GB of planes; above it blocks of at most 7 GB,
    whole 256-ray workgroups each.  (Sizes only: nothing is traced.)"""
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)          # 13 elements
    eng = ra.Engine()
    eng.set_option("block_rays", 0)       # automatic, whatever the environment
    g = ra.GeometricTrace(system, engine=eng)
    y, u = disc_bundle(1000, 12., 0., 1, P.DOUBLE_GAUSS_PUPIL_Z)
    g.rays_given(y, u)                                    # uploads the table
    for n, blocks in ((3_000_000, 1), (8_000_000, 1), (10_000_000, 2),
                      (12_500_000, 2), (20_000_000, 3), (30_000_000, 5)):
        eng.reserve(n)
        nb, bs, bts = eng.blocks()
        assert nb == blocks, (n, nb)
        assert nb*bs >= n and (nb == 1 or bs % 256 == 0)
        assert nb == 1 or 80*13*bs <= 7.1e9
    eng.close()
