"""Launch components that are one bit pattern across a 64-ray tile (the
direction of a collimated bundle, the origin of a bundle from an object
point, z = 0 of rays starting on a plane) are fetched once per wavefront:
noted per tile by the seed kernels, used by the trace from element 1, voided
by anything that rewrites row 0.  Same values -> same bits, always."""
import numpy as np
import pytest

import rayopt_amd as ra
from rayopt_amd import prescriptions as P
from rayopt_amd._lib import RT_Y, RT_U
from rayopt_amd.bundles import disc_bundle

pytestmark = pytest.mark.gpu


def rows_of(g):
    return [np.array(np.asarray(r[:])) for r in (g.y, g.u, g.i, g.t)]


def same(a, b):
    return all(np.array_equal(x, w, equal_nan=True) for x, w in zip(a, b))


def traced(system, y, u, **options):
    g = ra.GeometricTrace(system, **options)
    g.rays_given(y, u)
    g.propagate(clip=True)
    return g


@pytest.mark.parametrize("n", [64, 640, 1000, 100_032, 100_003])
def test_collimated_bundle_reads_a_third_and_gives_the_same_bits(n):
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    y, u = disc_bundle(n, 17., 7., 2, P.DOUBLE_GAUSS_PUPIL_Z)
    assert (u == u[0]).all() and (y[:, 2] == y[0, 2]).all()
    g = traced(system, y, u)
    uniform, tiles = g.engine.input_uniform()
    full = n//64                   # a tile with padding columns is read whole
    nb, bs, _ = g.engine.blocks()  # (one block unless the environment cuts)
    assert tiles == (nb*bs if nb > 1 else n + 63)//64
    assert uniform == [0, 0, full, full, full, full]
    plain = traced(system, y, u, uniform_input=0)
    assert plain.engine.input_uniform()[0] == [0]*6
    assert same(rows_of(g), rows_of(plain))
    # in pieces too (the notes are indexed with the window)
    g.rays_given(y, u)
    g.propagate(clip=True, chunks=3)
    assert same(rows_of(g), rows_of(plain))


def test_point_source_and_mixed_tiles():
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    rng = np.random.default_rng(5)
    n = 64*40
    y, u = disc_bundle(n, 12., 3., 1, P.DOUBLE_GAUSS_PUPIL_Z)
    # tiles 0-9: one origin (a point source), directions differ
    y[:640] = y[0]
    u[:640, :2] += rng.normal(size=(640, 2))*1e-3
    u[:640, 2] = np.sqrt(1 - np.square(u[:640, :2]).sum(1))
    # tiles 10-19: everything differs; 20-39: collimated, on a plane
    y[640:1280, 2] += rng.normal(size=640)*1e-3
    u[640:1280, :2] += rng.normal(size=(640, 2))*1e-3
    # NaN is a bit pattern like any other: tile 39 all-NaN in u0
    u[-64:, 0] = np.nan
    g = traced(system, y, u)
    uniform, tiles = g.engine.input_uniform()
    assert tiles == 40
    assert uniform[:3] == [10, 10, 30] and uniform[3] == 20
    assert same(rows_of(g), rows_of(traced(system, y, u, uniform_input=0)))


def test_whatever_rewrites_row_0_voids_the_notes():
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    n = 6400
    y, u = disc_bundle(n, 17., 7., 2, P.DOUBLE_GAUSS_PUPIL_Z)
    y2, u2 = disc_bundle(n, 12., -3., 9, P.DOUBLE_GAUSS_PUPIL_Z)
    u2[:, :2] += np.random.default_rng(1).normal(size=(n, 2))*1e-3
    want = rows_of(traced(system, y2, u2, uniform_input=0))
    # rows uploaded through the C ABI
    g = traced(system, y, u)
    assert any(g.engine.input_uniform()[0])
    g.engine.upload_row(RT_Y, 0, np.ascontiguousarray(y2.T))
    g.engine.upload_row(RT_U, 0, np.ascontiguousarray(u2.T))
    assert g.engine.input_uniform()[0] == [0]*6
    g.engine.trace(1, 0, True)
    assert np.array_equal(np.asarray(g.engine.download(RT_Y, 12, 13))[0].T,
                          want[0][-1], equal_nan=True)
    # a raw pointer to row 0 handed out: the caller may write through it
    g = traced(system, y, u)
    g.engine.device_ptr(RT_U, 0)
    assert g.engine.input_uniform()[0] == [0]*6
    # new rays: new notes
    g.rays_given(y2, u2)
    g.propagate(clip=True)
    assert same(rows_of(g), want)
    g.rays_given(y, u)
    assert g.engine.input_uniform()[0][3:] == [100, 100, 100]
    # a device-generated batch has none (it rebuilds its rays in registers)
    g.rays_fields(np.array([[0., .5]]), np.zeros((640, 2)),
                  P.DOUBLE_GAUSS_PUPIL_Z, 17.)
    g.propagate(clip=True)
    assert g.engine.input_uniform()[0] == [0]*6


def test_partial_traces_and_ray_groups_keep_their_bits():
    system = ra.system_from_yaml(P.COOKE % dict(
        air="air", sk16="SCHOTT-SK|N-SK16", f2="SCHOTT-F|N-F2"))
    y, u = disc_bundle(6400, 5.5, 5., 0)
    ls = [587.56e-9, 656.27e-9, 486.13e-9]
    a = ra.GeometricTrace(system)
    a.rays_given(y, u, l=ls)
    a.propagate(clip=True)
    b = ra.GeometricTrace(system, uniform_input=0)
    b.rays_given(y, u, l=ls)
    b.propagate(clip=True)
    assert same(rows_of(a), rows_of(b))
    # start > 1 seeds from another row: the notes (row 0) are not used
    a.propagate(start=3, clip=False)
    b.propagate(start=3, clip=False)
    assert same(rows_of(a), rows_of(b))


@pytest.mark.parametrize("n", [64*50, 64*50 + 17, 200_000])
def test_a_direction_s_third_component_is_rebuilt_where_it_is_the_completion(n):
    """u_z that is, bit for bit, sqrt(1 - (u_x^2 + u_y^2)) (what the reference
    writes for two-component directions, rayopt/geometric_trace.py:57-60) or
    sqrt((1 - u_x^2) - u_y^2) is not read by the trace but rebuilt with the
    same operations: noted per tile by the seed kernels, same bits always;
    a tile with one ray that is anything else (-u_z, a perturbed u_z, NaN)
    is read."""
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    rng = np.random.default_rng(n)
    y, u = disc_bundle(n, 15., 5., 3, P.DOUBLE_GAUSS_PUPIL_Z)
    u[:, :2] += 1e-3*rng.standard_normal((n, 2))
    full = n//64
    for form in (0, 1):
        if form:
            u[:, 2] = np.sqrt(1. - np.square(u[:, :2]).sum(1))
        else:
            u[:, 2] = np.sqrt(1. - u[:, 0]**2 - u[:, 1]**2)
        g = traced(system, y, u)
        assert g.engine.input_uniform()[0][3:] == [0, 0, 0]
        assert g.engine.input_completed() == full
        plain = traced(system, y, u, uniform_input=0)
        assert plain.engine.input_completed() == 0
        assert same(rows_of(g), rows_of(plain))
        assert np.array_equal(np.asarray(g.u[0]), u)
        g.rays_given(y, u)
        g.propagate(clip=True, chunks=3)
        assert same(rows_of(g), rows_of(plain))
    # tiles 3, 7 and 11 hold one ray each that is not the completion
    v = u.copy()
    v[3*64 + 5, 2] *= -1.
    v[7*64 + 63, 2] = np.nextafter(v[7*64 + 63, 2], 2.)
    v[11*64, :] = np.nan
    g = traced(system, y, v)
    assert g.engine.input_completed() == full - 3
    assert same(rows_of(g), rows_of(traced(system, y, v, uniform_input=0)))
    # a collimated bundle: u_z is uniform, not "completed"
    yc, uc = disc_bundle(6400, 17., 7., 2, P.DOUBLE_GAUSS_PUPIL_Z)
    assert traced(system, yc, uc).engine.input_completed() == 0
