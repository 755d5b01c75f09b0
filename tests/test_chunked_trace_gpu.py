"""rt_trace_chunk: the trace of a step in pieces of the ray batch (what a
multi-GPU job uses to overlap its gather with the trace).  All pieces together
must be the unchunked trace, bit for bit -- host-seeded and device-generated
batches (first trace and re-traces), partial ranges, kept-row subsets, ragged
sizes, chunks in any order -- and what cannot be chunked must say so."""
import numpy as np
import pytest

import rayopt_amd as ra
from rayopt_amd import prescriptions as P
from rayopt_amd.bundles import disc_bundle

pytestmark = pytest.mark.gpu


def rows_of(g):
    return [np.array(np.asarray(r[:])) for r in (g.y, g.u, g.i, g.t)]


@pytest.mark.parametrize("n", [1, 63, 257, 4099, 100_003])
@pytest.mark.parametrize("chunks", [2, 3, 8])
def test_host_seeded_chunks_equal_the_whole(n, chunks):
    system = ra.system_from_yaml(P.TORTURE)        # tilted: i rows stored
    y, u = disc_bundle(n, 12., 2., 5)
    g = ra.GeometricTrace(system)
    g.rays_given(y, u)
    g.propagate(clip=True)
    want = rows_of(g)
    h = ra.GeometricTrace(system)
    h.rays_given(y, u)
    seen = []
    h.propagate(clip=True, chunks=chunks,
                after_chunk=lambda k, q: seen.append((k, q)))
    assert seen == [(k, chunks) for k in range(chunks)]
    for a, b in zip(want, rows_of(h)):
        assert np.array_equal(a, b, equal_nan=True)
    # a partial range with a kept-row subset, chunks issued in reverse order
    h.rays_given(y, u)
    h.propagate(stop=4, clip=False)
    g.rays_given(y, u)
    g.propagate(stop=4, clip=False)
    g.propagate(start=4, clip=True, keep=[5, -1])
    eng = h.engine
    mask = np.zeros(len(system), dtype=np.uint8)
    mask[[5, len(system) - 1]] = 1
    mask[:4] = 1
    h._upload_table(4, len(system), h.n[3])
    eng.set_keep_rows(mask)
    for k in reversed(range(chunks)):
        eng.trace_chunk(4, len(system), True, k, chunks)
    for rows in (h.y, h.u, h.i, h.t):
        rows.invalidate(4, len(system))
    for name in "yuit":
        for j in (5, len(system) - 1):
            assert np.array_equal(np.asarray(getattr(g, name)[j]),
                                  np.asarray(getattr(h, name)[j]),
                                  equal_nan=True), (name, j)


def test_generated_batch_chunks_rebuild_the_rays():
    """A device-generated batch traced in chunks: the first trace builds row
    0 with the stand-alone generation kernel and every chunk rebuilds its
    rays in registers (the regen kernel with a ray offset)."""
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    fields = np.c_[np.zeros(3), (0., .5, 1.)]
    rng = np.random.default_rng(3)
    r, phi = np.sqrt(rng.random(20_000)), 2*np.pi*rng.random(20_000)
    yp = np.c_[r*np.cos(phi), r*np.sin(phi)]
    g = ra.GeometricTrace(system)
    g.rays_fields(fields, yp, P.DOUBLE_GAUSS_PUPIL_Z, 17.)
    g.propagate(clip=True)
    want = rows_of(g)
    for chunks in (2, 5):
        h = ra.GeometricTrace(system)
        h.rays_fields(fields, yp, P.DOUBLE_GAUSS_PUPIL_Z, 17.)
        h.propagate(clip=True, chunks=chunks)       # first trace, chunked
        for a, b in zip(want, rows_of(h)):
            assert np.array_equal(a, b, equal_nan=True)
        system[3].distance = system[3].distance     # touch: table re-sent
        h.propagate(clip=True, chunks=chunks)       # re-trace, chunked
        for a, b in zip(want, rows_of(h)):
            assert np.array_equal(a, b, equal_nan=True)


def test_what_cannot_be_chunked_says_so():
    system = ra.system_from_yaml(P.COOKE % P.COOKE_INDICES[587.56e-9])
    y, u = disc_bundle(640, 5., 2., 1)
    g = ra.GeometricTrace(system)
    g.rays_given(y, u, l=[587.56e-9, 656.27e-9])    # two ray groups
    with pytest.raises(ra.EngineError, match="ray groups"):
        g.propagate(clip=True, chunks=2)
    g.propagate(clip=True)                          # unchunked: fine
    with pytest.raises(ra.EngineError, match="chunk"):
        g.engine.trace_chunk(1, 0, True, 3, 3)
    assert g.engine.chunk_bounds(1000, 0, 1) == (0, 1000)


def test_rows_are_not_read_whole_in_the_middle_of_a_step_in_pieces():
    """Until the last piece of a step has been traced the rows hold new and
    old columns side by side: every whole-row reader says so instead of
    handing out the mixture; the pieces themselves can be issued in any order
    and a complete step (or an unchunked trace) opens the rows again."""
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    L = len(system)
    y, u = disc_bundle(5000, 12., 2., 5)
    g = ra.GeometricTrace(system)
    g.rays_given(y, u)
    g.propagate(clip=True)
    want = np.array(g.y[-1])
    eng = g.engine
    system[4].distance *= 1.001             # the next trace changes the rows
    g._upload_table(1, L, g.n[0])
    eng.trace_chunk(1, L, True, 2, 3)
    for call in (lambda: eng.download(0, L - 1, L), lambda: eng.rms(L - 1),
                 lambda: eng.row_rmax(L - 1), lambda: eng.device_ptr(0, 1),
                 lambda: eng.refocus_shift(L - 1)):
        with pytest.raises(ra.EngineError, match="pieces of the current step"):
            call()
    eng.trace_chunk(1, L, True, 0, 3)
    eng.trace_chunk(1, L, True, 0, 3)       # a piece twice: still one piece
    with pytest.raises(ra.EngineError, match="2 of 3 pieces"):
        eng.download(0, L - 1, L)
    eng.trace_chunk(1, L, True, 1, 3)
    got = eng.download(0, L - 1, L)         # complete: readable again
    h = ra.GeometricTrace(system)
    h.rays_given(y, u)
    h.propagate(clip=True)
    assert np.array_equal(got[0].T, np.array(h.y[-1]), equal_nan=True)
    assert not np.array_equal(got[0].T, want, equal_nan=True)
    # an unchunked trace ends an unfinished step in pieces as well
    eng.trace_chunk(1, L, True, 0, 4)
    with pytest.raises(ra.EngineError):
        eng.rms(L - 1)
    eng.trace(1, L, True)
    assert np.isfinite(eng.rms(L - 1))


def test_a_new_batch_ends_a_step_in_pieces():
    """a step in pieces that is abandoned half way (after_chunk
    raised, say) must not be completed by pieces of the NEXT batch.  Whatever
    replaces the rays, a row or the table starts the count again: the rows
    read whole at once (they hold what the reference's arrays would -- stale
    rows of the old batch), and pieces of the new batch count from zero."""
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    L = len(system)
    y, u = disc_bundle(5000, 12., 2., 5)
    y2, u2 = disc_bundle(5000, 9., 1., 6)
    g = ra.GeometricTrace(system)
    g.rays_given(y, u)
    g.propagate(clip=True)
    eng = g.engine
    eng.trace_chunk(1, L, True, 0, 3)
    with pytest.raises(ra.EngineError, match="1 of 3 pieces"):
        eng.download(0, L - 1, L)
    g.rays_given(y2, u2)                    # the step is abandoned here
    eng.download(0, L - 1, L)               # no step in progress any more
    g._upload_table(1, L, g.n[0])
    eng.trace_chunk(1, L, True, 1, 3)
    eng.trace_chunk(1, L, True, 2, 3)
    # before the fix these two completed the old step and the image row was
    # handed out with the first third of its columns from the old batch
    with pytest.raises(ra.EngineError, match="2 of 3 pieces"):
        eng.download(0, L - 1, L)
    eng.trace_chunk(1, L, True, 0, 3)
    got = eng.download(0, L - 1, L)
    h = ra.GeometricTrace(system)
    h.rays_given(y2, u2)
    h.propagate(clip=True)
    assert np.array_equal(got[0].T, np.array(h.y[-1]), equal_nan=True)
    # a row upload and a changed table end a step as well
    for spoil in (lambda: eng.upload_row(0, 0, np.ascontiguousarray(y2.T)),
                  lambda: (system[4].__setattr__(
                      "distance", system[4].distance*1.001),
                      g._upload_table(1, L, g.n[0]))):
        eng.trace_chunk(1, L, True, 0, 2)
        with pytest.raises(ra.EngineError):
            eng.rms(L - 1)
        spoil()
        eng.trace_chunk(1, L, True, 1, 2)
        with pytest.raises(ra.EngineError, match="1 of 2 pieces"):
            eng.rms(L - 1)
        eng.trace_chunk(1, L, True, 0, 2)
        assert np.isfinite(eng.rms(L - 1))
