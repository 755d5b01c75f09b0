"""rt_newton_census: how the per-wavefront trip count of the asphere
iteration fits a batch (north_star: "wavefront ballots for the asphere
iteration"), counted on the device by a march that stores nothing, against
the iterate counts of the numpy restatement of the reference's solver
(rayopt/elements.py:333-349)."""
import os
import sys

import numpy as np
import pytest

import rayopt_amd as ra
from rayopt_amd import prescriptions as P
from rayopt_amd.pack import pack_system
from oracle import trace_numpy as tn

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import digest_cases as dc                           # noqa: E402

pytestmark = pytest.mark.gpu


def oracle_census(system, y, u, l, clip):
    table, _ = pack_system(system, l, system.refractive_index(l, 0))
    tn.ITERATES = []
    try:
        tn.propagate(table, y, u, clip=clip)
        per_surface = tn.ITERATES
    finally:
        tn.ITERATES = None
    slots = iterates = trips = solves = 0
    pad = -len(y) % 64
    for it, dead in per_surface:
        w = np.concatenate([it, np.zeros(pad, dtype=it.dtype)]).reshape(-1, 64)
        d = np.concatenate([dead, np.ones(pad, dtype=bool)]).reshape(-1, 64)
        # a wavefront whose rays are all dead skips the element (rt_march);
        # in the others a dead ray goes through its one NaN iterate
        runs = ~d.all(1)
        t = np.where(runs, w.max(1), 0)
        trips += int(t.sum())
        slots += 64*int(t.sum())
        iterates += int(w[runs].sum())
        solves += int(runs.sum())
    return {"lane_slots": slots, "iterates": iterates, "wave_trips": trips,
            "wave_solves": solves}


@pytest.mark.parametrize("radius,clip,n", [(.6, True, 64*700 + 17),
                                           (1.6, True, 64*500),
                                           (.6, False, 5000)])
def test_census_matches_the_solver_ray_by_ray(radius, clip, n):
    """Exact arithmetic: the device's counts ARE the reference solver's, ray
    by ray and wavefront by wavefront -- also where a wide bundle is clipped
    on the way: a ray that arrives dead is retired after its first (NaN)
    iterate and no longer holds its wavefront for five trips.  Default arithmetic: the same
    counts up to the rare ray whose step lies within rounding of 1e-7."""
    system = ra.system_from_yaml(P.ASPHERE_PHONE)
    y, u = dc.bundle(n, radius, 10., 4)
    y[:, 1] -= .5*np.tan(np.radians(10.))
    l = system.wavelengths[0]
    want = oracle_census(system, y, u, l, clip)
    assert want["wave_solves"] > 0
    for opts in ({"exact_asphere": 1}, {}):
        g = ra.GeometricTrace(system, **opts)
        g.rays_given(y, u, l)
        with pytest.raises(ra.EngineError, match="trace the batch"):
            g.engine.newton_census(clip)
        g.propagate(clip=clip)
        got = g.engine.newton_census(clip)
        if opts:
            assert {k: got[k] for k in want} == want
        else:
            for k in want:
                assert got[k] == pytest.approx(want[k], rel=2e-3), k
        assert got["lane_utilisation"] == pytest.approx(
            got["iterates"]/got["lane_slots"])
        # the census changes nothing: the rows are those of the trace
        before = np.array(g.y[-1])
        g.engine.newton_census(clip)
        assert np.array_equal(before, np.array(g.engine.download(
            0, len(system) - 1, len(system))[0].T), equal_nan=True)
    if radius > 1. and clip:
        # some rays died on the way and wavefronts went on without them
        dead = np.isnan(np.asarray(g.u[-1])[:, 0]).mean()
        assert 0 < dead < 1
        assert want["wave_trips"] < 5*want["wave_solves"]


def test_census_of_a_system_without_aspheres_is_empty():
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    y, u = ra.bundles.disc_bundle(10_000, 12., 3., 1, P.DOUBLE_GAUSS_PUPIL_Z)
    g = ra.GeometricTrace(system)
    g.rays_given(y, u)
    g.propagate(clip=True)
    got = g.engine.newton_census(True)
    assert got["lane_slots"] == 0 and got["lane_utilisation"] is None
