"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the consumers of a trace.

numpy restatements of the O(N) post-processing GeometricTrace offers on the
result arrays (quartiq/rayopt), used to check the device-side reductions
``rt_rms``, ``rt_refocus_shift`` and ``rt_opd_rays``.  Same whole-array numpy
operations as the reference; pinned against the unmodified reference by the
``consumers_*.npz`` goldens (tests/golden/make_golden.py) and, when
/root/reference is present, live (tests/test_consumers.py).  Never imported
by the product.
"""
import numpy as np


def rms(y_row, w=None, ref=None):
    """GeometricTrace.rms (rayopt/geometric_trace.py:171-183); ``y_row`` is
    ``trace.y[i]`` (N,3)."""
    y = y_row[:, :2]
    y0 = y.mean(0) if ref is None else y[ref]
    r = np.square(y - y0).sum(1)
    if w is None:
        w = np.ones_like(r)/r.shape[0]
    return np.sqrt((r*w).sum())


def spot_stats(y_row, group_rays, w=None):
    """Per-bundle spot statistics for a batch of contiguous bundles of
    ``group_rays`` rays -- ``rms`` above applied to every bundle on its own
    with its weights normalised, restricted to the rays that arrived
    (finite intercept); checks ``rt_spot_stats``.  Rows: count, centroid x,
    centroid y, sum(w d^2)/sum(w), max d^2, sum(w)."""
    n = y_row.shape[0]
    groups = n//group_rays
    assert groups*group_rays == n
    out = np.empty((groups, 6))
    for g in range(groups):
        sl = slice(g*group_rays, (g + 1)*group_rays)
        y = y_row[sl, :2]
        wg = np.ones(group_rays) if w is None else np.asarray(w)[sl]
        good = np.all(np.isfinite(y), axis=1)
        y, wg = y[good], wg[good]
        if not len(y):
            out[g] = 0., np.nan, np.nan, np.nan, np.nan, 0.
            continue
        y0 = y.mean(0)
        r = np.square(y - y0).sum(1)
        out[g] = len(y), y0[0], y0[1], (r*wg).sum()/wg.sum(), r.max(), \
            wg.sum()
    return out


def row_stats(y_row, group_rays, w=None, ref=None):
    """What ``rt_row_stats`` returns per bundle, from the reference's own
    formulas: ``rms`` above about the mean and about the bundle's ray ``ref``
    (rayopt/geometric_trace.py:171-183), the largest distance from the axis
    (``resize``, :185-193), over the rays that arrived, weights normalised.
    Rows: count, sum w, mean x, mean y, sum(w d^2)/sum(w) about the mean,
    the same about ray ``ref`` (NaN: none, or it did not arrive), max(x^2 +
    y^2), weighted centroid x, y."""
    n = y_row.shape[0]
    groups = n//group_rays
    assert groups*group_rays == n
    out = np.full((groups, 9), np.nan)
    for g in range(groups):
        sl = slice(g*group_rays, (g + 1)*group_rays)
        y = y_row[sl, :2]
        wg = np.ones(group_rays) if w is None else np.asarray(w)[sl]
        good = np.all(np.isfinite(y), axis=1)
        yref = y[ref] if ref is not None and ref >= 0 else None
        y, wg = y[good], wg[good]
        out[g, :2] = len(y), wg.sum()
        if not len(y):
            continue
        y0 = y.mean(0)
        out[g, 2:4] = y0
        out[g, 4] = (np.square(y - y0).sum(1)*wg).sum()/wg.sum()
        if yref is not None and np.isfinite(yref).all():
            out[g, 5] = (np.square(y - yref).sum(1)*wg).sum()/wg.sum()
        out[g, 6] = np.square(y).sum(1).max()
        out[g, 7:9] = (y*wg[:, None]).sum(0)/wg.sum()
    return out


def refocus_shift(y_row, i_row, w=None):
    """The shift ``t`` GeometricTrace.refocus adds to ``system[at].distance``
    (rayopt/geometric_trace.py:82-97)."""
    y = y_row[:, :2]
    u = i_row[:, :2]/i_row[:, 2:]          # tanarcsin, rayopt/utils.py:47-50
    good = np.all(np.isfinite(u), axis=1)
    y, u = y[good], u[good]
    w = w[good] if w is not None else np.ones(y.shape[0])
    y = y - y.mean(0)
    u = u - u.mean(0)
    wy = (w[:, None]*y).ravel()
    wu = (w[:, None]*u).ravel()
    u = u.ravel()
    return -np.dot(wy, u)/np.dot(wu, u)


def sphere_intercept(curvature, y, u):
    """Spheroid(curvature=c).intercept, k=0 (rayopt/elements.py:477-501)."""
    c = curvature
    if c == 0:
        return -y[:, 2]/u[:, 2]
    uy = (u*y).sum(1)
    yy = np.square(y).sum(1)
    d = c*uy - u[:, 2]
    e = c*1.
    f = c*yy - 2*y[:, 2]
    g = np.sqrt(np.square(d) - e*f)
    return -(d + g)/e


def opd_rays(Y, U, T, n, ref, origins, frames, finite, radius, lscale,
             after=-2, image=-1):
    """GeometricTrace.opd up to (not including) the resampling
    (rayopt/geometric_trace.py:101-131).  ``Y,U`` (L,N,3), ``T`` (L,N), ``n``
    (L,), ``frames[j]`` = rot_normal of element j or None, ``lscale`` =
    l/system.scale.  Returns x, y, t per ray."""
    def from_normal(j, v):
        return v if frames[j] is None else np.dot(v, frames[j])

    def to_normal(j, v):
        return v if frames[j] is None else np.dot(v, frames[j].T)

    t = (T[:after + 1] - T[:after + 1, (ref,)]).sum(0)
    if not finite:
        tj = np.dot(U[0, ref], (Y[0, ref] - Y[0]).T)
        t -= tj*n[0]
    y = from_normal(after, Y[after])
    y = y + (origins[after] - origins[image])
    y = to_normal(image, y) - Y[image, ref]
    u = to_normal(image, from_normal(after, U[after]))
    y[:, 2] += radius
    ti = sphere_intercept(1./radius, y, u)
    t += (ti - ti[ref])*n[after]
    t = -t/lscale
    py = y + ti[:, None]*u
    py[:, 2] -= radius
    py -= py[ref]
    x, yy, z = py.T
    return x, yy, t
